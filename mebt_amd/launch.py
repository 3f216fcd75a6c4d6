"""One command starts a multi-GPU job (reference: `python train_transformer.py --gpus 8`, train_transformer.py:39-46 — Lightning
spawns one process per device from the single command).  `bench.py --gpus N` / `python -m mebt_amd.train --gpus N` called
WITHOUT a torch.distributed environment re-launch themselves as N ranks: the parent decides before anything touches the GPU,
starts `python -m torch.distributed.run` as a CHILD process, relays rank 0's stdout and exits with the child's return code
(never an exec: a process that has initialised the GPU must not replace itself, and this one never initialises it anyway)."""
import os
import socket
import subprocess
import sys


def in_distributed_env():
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_command(n_ranks, script, argv, port=None, module=False):
    """the torch.distributed.run command line for `n_ranks` ranks of `script argv...` on this node"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(n_ranks)}",
           "--master-addr", "127.0.0.1", "--master-port", str(port or free_port())]
    cmd += (["-m", script] if module else [script]) + list(argv)
    return cmd


def spawn_ranks_if_needed(n_ranks, script, argv, module=False):
    """Returns None when this process should run the job itself (one rank requested, or already a rank of a distributed
    launch).  Otherwise runs the N-rank job as a child process and returns its exit code; the caller must `sys.exit` with it
    without touching the GPU."""
    if int(n_ranks) <= 1 or in_distributed_env():
        return None
    import torch                                  # importing torch does not initialise the GPU
    assert not torch.cuda.is_initialized(), "the launcher parent must not have touched the GPU"
    override = os.environ.get("MEBT_LAUNCH_CHILD")          # tests: run this command instead (shlex-split), same relay logic
    if override:
        import shlex
        cmd = shlex.split(override)
    else:
        cmd = launch_command(n_ranks, script, argv, module=module)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (RCCL / tensor sharing between the ranks)
    env.setdefault("OMP_NUM_THREADS", "4")
    print(f"[launch] starting {n_ranks} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE)
    for line in proc.stdout:                      # rank 0's result line(s), relayed as they come
        sys.stdout.buffer.write(line)
        sys.stdout.buffer.flush()
    rc = proc.wait()
    print(f"[launch] child exited with {rc}; parent cuda_initialized={torch.cuda.is_initialized()}", file=sys.stderr, flush=True)
    return rc


def csrc_fingerprint():
    """sha256 over the kernel / engine sources (mebt_amd/csrc/*.{hip,h,inc,cpp}, sorted by name): stored with a committed PMC
    traffic profile so that bench.py can tell a stale figure (tools/pmc_traffic.py writes it, bench.py compares it)"""
    import glob
    import hashlib
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(root, "*"))):
        if f.endswith((".hip", ".h", ".inc", ".cpp")):
            h.update(os.path.basename(f).encode())
            with open(f, "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()
