"""ctypes binding of libmebt_hip.so (C ABI: include/mebt_hip.h).

There is deliberately NO fallback: if the HIP library is missing or fails to load, importing the
compute path raises.  PyTorch is used only for device memory, streams and torch.distributed.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MEBT_HIP_LIB: another BUILD of the same library (tools/ab_bench.sh times two builds on one box); same ABI, same failure if absent
LIB_PATH = os.environ.get("MEBT_HIP_LIB") or os.path.join(_HERE, "lib", "libmebt_hip.so")

SHIPPED_TUNE = os.path.join(_HERE, "tune", "gfx950.txt")

MEBT_MAX_LAYERS = 128
F32, BF16, F16 = 0, 1, 2
EPI_NONE, EPI_GELU, EPI_RESID, EPI_GELU_BWD = 0, 1, 2, 3
MODE_IDS = {"latent_enc": 0, "latent_self": 1, "latent_dec": 2, "lt2l": 3, "maskgit": 4}

c_i32, c_i64, c_f32, c_vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p


class ModelDesc(C.Structure):
    _fields_ = [("n_layer", c_i32), ("n_head", c_i32), ("n_embd", c_i32), ("vocab", c_i32),
                ("n_latent", c_i32), ("block_size", c_i32), ("dtype", c_i32),
                ("modes", c_i32 * MEBT_MAX_LAYERS), ("label_smoothing", c_f32),
                ("embd_pdrop", c_f32), ("resid_pdrop", c_f32), ("attn_pdrop", c_f32)]


# name -> (restype, argtypes); every symbol declared in include/mebt_hip.h
PROTOTYPES = {
    "mebt_last_error": (C.c_char_p, []),
    "mebt_abi_version": (c_i32, []),
    "mebt_model_create": (c_i32, [C.POINTER(ModelDesc), C.POINTER(c_vp)]),
    "mebt_model_destroy": (None, [c_vp]),
    "mebt_model_param_counts": (c_i32, [c_vp, C.POINTER(c_i64), C.POINTER(c_i64)]),
    "mebt_model_bind": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "mebt_model_sync_lowp": (c_i32, [c_vp, c_vp]),
    "mebt_workspace_bytes": (c_i64, [c_vp, c_i32, c_i32, c_i32, c_i32]),
    "mebt_forward": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, C.c_uint64, c_vp]),
    "mebt_kvcache_bytes": (c_i64, [c_vp, c_i32, c_i32]),
    "mebt_forward_kvcache": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_i32, c_vp]),
    "mebt_gpt_forward": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "mebt_gpt_forward_train": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, C.c_uint64, c_vp]),
    "mebt_gpt_backward": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "mebt_loss": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp]),
    "mebt_loss_with_grad": (c_i32, [c_vp, c_vp, c_vp, c_vp, C.c_float, c_vp]),
    "mebt_backward_head": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_f32, c_vp]),
    "mebt_backward_head_dlogits": (c_i32, [c_vp, c_vp, c_vp, c_vp]),
    "mebt_backward_layers": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_vp]),
    "mebt_backward_embed": (c_i32, [c_vp, c_vp, c_vp]),
    "mebt_adamw_step": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_f32, c_f32, c_f32, c_f32, c_f32, c_i32, c_f32, c_vp]),
    "mebt_adamw_range": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_f32, c_f32, c_f32, c_f32, c_f32, c_i32, c_f32, c_i32, c_i32, c_i32, c_vp]),
    "mebt_adamw_slice": (c_i32, [c_vp, c_i32, c_i64, c_i64, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_f32, c_f32, c_f32, c_f32, c_f32, c_i32, c_f32, c_vp]),
    "mebt_adamw_slice_pieces": (c_i32, [c_vp, c_i32, c_i64, c_i64, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_f32, c_f32, c_f32, c_f32, c_f32, c_i32, c_f32, c_vp]),
    "mebt_model_set_fused_adamw": (c_i32, [c_vp, c_vp, c_vp, c_f32, c_f32, c_f32, c_f32, c_f32, c_i32, c_f32]),
    "mebt_model_set_grad_accumulate": (c_i32, [c_vp, c_i32]),
    "mebt_model_bind_wire_grads": (c_i32, [c_vp, c_vp]),
    "mebt_model_set_forward_waits": (c_i32, [c_vp, c_i32, C.POINTER(C.c_int32), C.POINTER(C.c_void_p)]),
    "mebt_op_gemm": (c_i32, [c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp] + [c_i32] * 13 + [c_vp]),
    "mebt_op_layernorm_fwd": (c_i32, [c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp]),
    "mebt_op_layernorm_bwd": (c_i32, [c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp]),
    "mebt_op_attention_fwd": (c_i32, [c_i32, c_vp, c_vp, c_vp, c_vp, c_vp] + [c_i32] * 10 + [c_vp]),
    "mebt_op_attention_bwd": (c_i32, [c_i32] + [c_vp] * 10 + [c_i32] * 10 + [c_vp]),
    "mebt_op_embed_fwd": (c_i32, [c_i32] + [c_vp] * 10 + [c_i32] * 8 + [c_vp]),
    "mebt_op_sample": (c_i32, [c_vp, c_vp, c_f32, c_i32, c_f32, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp]),
    "mebt_op_sample_seeded": (c_i32, [c_vp, C.c_uint64, c_f32, c_i32, c_f32, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp]),
    "mebt_op_sample_scatter": (c_i32, [c_vp, c_vp, C.c_uint64, c_f32, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "mebt_op_sample_lp": (c_i32, [c_vp, c_i32, c_vp, C.c_uint64, c_f32, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "mebt_op_wgrad_grouped": (c_i32, [c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_f32, c_f32, c_f32,
                                      c_f32, c_f32, c_i32, c_f32, c_vp]),
    "mebt_op_topk_threshold": (c_i32, [c_vp, c_i32, c_vp, c_vp, c_i32, c_i32, c_vp]),
    "mebt_op_scatter_ids": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "mebt_op_next_mask": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_f32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp]),
    "mebt_op_cast_bf16": (c_i32, [c_vp, c_vp, c_i64, c_vp]),
    "mebt_op_conv3d": (c_i32, [c_i32, c_vp, c_i32, c_vp]),
    "mebt_op_groupnorm_silu": (c_i32, [c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i64, c_i32, c_vp]),
    "mebt_op_codebook_argmin": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "mebt_op_codebook_argmin_filtered": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "mebt_op_embedding_rows": (c_i32, [c_i32, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_vp]),
    "mebt_op_cast_f16": (c_i32, [c_vp, c_vp, c_i64, c_vp]),
    "mebt_debug_dropout_mask": (c_i32, [C.c_uint64, C.c_uint32, c_f32, c_i64, c_vp, c_vp]),
    "mebt_debug_clock_probe": (c_i32, [c_vp, C.c_uint64, c_vp]),
    "mebt_debug_side_stream": (None, [c_vp, c_i32]),
    "mebt_debug_set_side_stream": (None, [c_vp, c_vp]),
    "mebt_debug_gemm_tile": (None, [c_i32, c_i32]),
    "mebt_debug_gemm_variant": (None, [c_i32]),
    "mebt_debug_grouped_stages": (None, [c_i32]),
    "mebt_debug_gemm_scratch": (None, [c_vp, c_i64]),
    "mebt_debug_gemm_stamps": (None, [c_vp]),
    "mebt_debug_grouped_config": (None, [c_i32, c_i32, c_i32]),
    "mebt_gemm_autotune": (None, [c_i32]),
    "mebt_gemm_autotune_enabled": (c_i32, []),
    "mebt_gemm_tune_export": (c_i64, [C.c_char_p, c_i64]),
    "mebt_gemm_tune_import": (c_i32, [C.c_char_p, c_i32]),
    "mebt_gemm_tune_alternatives": (c_i64, [C.c_char_p, c_i64]),
    "mebt_profile_enable": (c_i32, [c_i32]),
    "mebt_profile_read": (c_i32, [c_i32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "mebt_profile_dump": (c_i64, [C.c_char_p, c_i64]),
    "mebt_profile_read_waits": (c_i32, [c_i32, C.POINTER(C.c_int32), C.POINTER(C.c_double)]),
}

_lib = None


def load():
    """Load libmebt_hip.so and bind every prototype.  Raises if the library is absent: the product
    path has no CPU / eager fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C mebt_amd/csrc`). mebt_amd has no fallback path without its HIP library.")
    # torch first: it bundles a HIP runtime of its own, and a process that loads /opt/rocm's libamdhip64 (through this library) BEFORE torch's
    # ends up with two runtimes and "no ROCm-capable device" at the first HIP call (seen with `build(); smoke()` in one process, round 6)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    # the shipped table of tuned GEMM configurations (choices measured on MI355X for the shipped configs' signatures): merged
    # WITHOUT overwriting, so MEBT_GEMM_TUNE_CACHE and in-situ tuning keep precedence and unseen signatures still tune at their
    # first launch.  MEBT_GEMM_TUNE_SHIPPED=0 starts from an empty table (fresh-tuning runs).
    if os.environ.get("MEBT_GEMM_TUNE_SHIPPED", "1") != "0" and os.path.exists(SHIPPED_TUNE):
        with open(SHIPPED_TUNE, "rb") as f:
            lib.mebt_gemm_tune_import(f.read(), 0)
    return lib


def tune_table_text():
    """the process-wide GEMM configuration table as text (include/mebt_hip.h: mebt_gemm_tune_export)"""
    lib = load()
    n = lib.mebt_gemm_tune_export(None, 0)
    buf = C.create_string_buffer(int(n))
    lib.mebt_gemm_tune_export(buf, n)
    return buf.value.decode()


def tune_table_merge(text, overwrite=True, replace=False):
    return int(load().mebt_gemm_tune_import(text.encode() if isinstance(text, str) else text, 2 if replace else (1 if overwrite else 0)))


class MebtError(RuntimeError):
    pass


def check(status):
    """Non-zero C status -> exception carrying mebt_last_error() (reference convention: Python
    exceptions / asserts)."""
    if status != 0:
        msg = load().mebt_last_error()
        raise MebtError(f"libmebt_hip status {status}: {msg.decode() if msg else '?'}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def cur_stream():
    import torch
    return torch.cuda.current_stream().cuda_stream
