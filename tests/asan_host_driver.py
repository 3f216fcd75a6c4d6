"""Driver of the host-side sanitizer test (tests/test_host_cpu.py::test_host_layer_under_asan): runs in a subprocess under
LD_PRELOAD=<clang asan runtime> with MEBT_HOST_ONLY=1 and walks the C-ABI's host-only paths of libmebt_hip_asan.so
(`make -C mebt_amd/csrc asan`): model creation / validation for every shipped geometry and block-mode mix, the flat-parameter
offset tables, the workspace carving at ragged (NC, NT) incl. NC = 0 and block 8192, argument validation of the operator
entry points, the error-string plumbing, create / destroy cycles.  No GPU is touched; ASan / UBSan abort the process on a finding."""
import ctypes as C
import sys

sys.path.insert(0, sys.argv[2])
from mebt_amd import _lib

_lib.LIB_PATH = sys.argv[1]
lib = _lib.load()
assert lib.mebt_abi_version() >= 1


def desc(n_layer, n_head, d, vocab, ns, block, dtype, modes, drop=0.1):
    m = _lib.ModelDesc()
    m.n_layer, m.n_head, m.n_embd, m.vocab, m.n_latent, m.block_size, m.dtype = n_layer, n_head, d, vocab, ns, block, dtype
    for i, name in enumerate(modes):
        m.modes[i] = _lib.MODE_IDS[name]
    m.label_smoothing, m.embd_pdrop, m.resid_pdrop, m.attn_pdrop = 0.0, drop, drop, drop
    return m


sky = ["latent_enc", "latent_self"] * 6 + ["latent_enc"] + ["latent_dec", "lt2l"] * 5 + ["latent_dec"]
good = [desc(24, 16, 1024, 16384, 256, 1024, _lib.BF16, sky), desc(24, 16, 1024, 16384, 256, 8192, _lib.BF16, sky, 0.0),
        desc(4, 4, 256, 16384, 64, 256, _lib.F32, ["latent_enc", "latent_self", "latent_dec", "lt2l"]),
        desc(3, 2, 64, 16384, 8, 32, _lib.BF16, ["latent_enc", "latent_dec", "maskgit"]),
        desc(2, 2, 128, 16384, 0, 64, _lib.F32, ["maskgit", "maskgit"])]
n_ok = 0
for d in good:
    for rep in range(3):                                     # create / destroy cycles
        h = C.c_void_p()
        rc = lib.mebt_model_create(C.byref(d), C.byref(h))
        assert rc == 0, lib.mebt_last_error()
        nw, np_ = C.c_int64(), C.c_int64()
        assert lib.mebt_model_param_counts(h, C.byref(nw), C.byref(np_)) == 0 and nw.value > 0 and np_.value > 0
        N = d.block_size
        for B in (1, 3, 6):
            for NC, NT in ((0, N), (N - 1, 1), (N // 2, N // 2), (N // 3, N - N // 3), (min(N, 7936), min(N, 256))):
                for training in (0, 1):
                    by = lib.mebt_workspace_bytes(h, B, NC, NT, training)
                    assert by > 0, (B, NC, NT, training)
        assert lib.mebt_workspace_bytes(h, -1, 0, 1, 0) == -1
        # launches without a bound model / workspace must fail with a status and a message, not crash
        assert lib.mebt_forward(h, None, 0, 1, N, 0, N, None, None, None, None, 0, 0, None) != 0 and lib.mebt_last_error()
        assert lib.mebt_backward_layers(h, None, 0, 0, None) != 0
        assert lib.mebt_adamw_range(h, None, None, None, None, 1e-3, 0.9, 0.95, 1e-8, 0.01, 1, 1.0, 4, 0, 0, None) != 0
        lib.mebt_model_destroy(h)
        n_ok += 1

bad = [desc(0, 4, 256, 16384, 64, 256, _lib.F32, []), desc(4, 3, 256, 16384, 64, 256, _lib.F32, ["latent_enc"] * 4),
       desc(4, 4, 200, 16384, 64, 256, _lib.F32, ["latent_enc"] * 4), desc(4, 4, 256, 16383, 64, 256, _lib.F32, ["latent_enc"] * 4),
       desc(4, 4, 256, 16384, 0, 256, _lib.F32, ["latent_enc"] * 4), desc(4, 4, 256, 16384, 64, 256, 7, ["latent_enc"] * 4),
       desc(4, 4, 256, 16384, 64, 256, _lib.F32, ["latent_enc"] * 4, drop=1.0)]
for d in bad:
    h = C.c_void_p()
    assert lib.mebt_model_create(C.byref(d), C.byref(h)) != 0 and len(lib.mebt_last_error()) > 10
d = good[2]
d.modes[1] = 9
h = C.c_void_p()
assert lib.mebt_model_create(C.byref(d), C.byref(h)) != 0
assert lib.mebt_model_create(None, C.byref(h)) != 0
# operator entry points: null pointers / bad shapes are rejected before anything is launched
assert lib.mebt_op_gemm(_lib.BF16, None, None, None, None, None, None, 128, 128, 64, 64, 64, 128, 128, 1, 1, 0, 0, 0, 1, None) != 0
assert lib.mebt_op_sample(None, None, 1.0, 0, 0.0, None, None, None, 4, 16384, None) != 0
assert lib.mebt_op_topk_threshold(None, 5, None, None, 4, 16384, None) != 0
print(f"asan host driver: {n_ok} model handles, {len(bad) + 2} rejected descriptors, no sanitizer finding")
