#!/usr/bin/env python3
"""What would a last-arriver split-K have to beat?  Fresh in-situ tuning of one Sky-16f train step with every
candidate logged (MEBT_GEMM_TUNE_LOG=2, shipped table off); for each single-product signature that has split-K candidates this prints
the best unsplit candidate, the best split-K candidate (fp32 slabs + splitk_reduce_kernel) and that candidate minus the reduce
kernel's own time — the floor of a variant whose last-arriving workgroup reduces in place of a second launch (it still writes and
re-reads the partial tiles).   GPU box: python tools/splitk_candidates.py"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
env = dict(os.environ, MEBT_GEMM_TUNE_LOG="2", MEBT_GEMM_TUNE_SHIPPED="0")
env.pop("MEBT_GEMM_TUNE_CACHE", None)
r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--secondary", "none", "--no-cpu-baseline"],
                   env=env, capture_output=True, text=True, cwd=ROOT)
cands, out = [], []
for ln in r.stderr.splitlines():
    m = re.match(r"\s+cand (\d+)x(\d+) ring (\d) ?(.*?): ([\d.]+) us", ln)
    if m:
        cands.append((m.group(1) + "x" + m.group(2) + " ring " + m.group(3) + (" " + m.group(4) if m.group(4) else ""), float(m.group(5))))
        continue
    m = re.match(r"\[mebt gemm autotune\] M=(\d+) N=(\d+) K=(\d+) a_kc=(\d) b_kc=(\d) epi=(\d) c_f32=(\d) -> (.*)", ln)
    if m:
        M, N, K = int(m.group(1)), int(m.group(2)), int(m.group(3))
        sk = [c for c in cands if "split-K" in c[0]]
        un = [c for c in cands if "split-K" not in c[0]]
        if sk and un:
            b_un, b_sk = min(un, key=lambda c: c[1]), min(sk, key=lambda c: c[1])
            S = 4 if "split-K 4" in b_sk[0] else 2
            # the reduce launch alone: S fp32 slabs read + one bf16 output written at ~4 TB/s, never below the ~3.5 us launch floor
            red = max(3.5, (S * 4 + 2) * M * N / 4.0e6)
            out.append((M, N, K, int(m.group(4)), int(m.group(5)), int(m.group(6)), b_un, b_sk, red))
        cands = []
    elif ln.startswith("[mebt gemm autotune]"):
        cands = []
print(f"{'product (M x N x K, layouts, epilogue)':42s} {'best unsplit candidate':34s} {'best split-K candidate':38s} {'- reduce launch':>15s}   verdict")
for M, N, K, ak, bk, epi, b_un, b_sk, red in out:
    floor = b_sk[1] - red
    print(f"{M:5d} x {N:5d} x {K:5d} a_kc={ak} b_kc={bk} epi={epi}   {b_un[0]:24s} {b_un[1]:6.1f} us   {b_sk[0]:28s} {b_sk[1]:6.1f} us   {floor:8.1f} us      {'unsplit wins' if b_un[1] <= floor else 'a fused reduce could win by %.1f us' % (b_un[1] - floor)}")
if not out:
    print(r.stderr[-3000:])
