"""ctypes binding of libcodae_hip.so (include/codae_hip.h).

The shared object is built in-tree by `__graft_entry__.build()` /
`make -C mui-deepautoencoder_amd/csrc`.  There is no CPU fallback: if the
library is missing, every compute entry point of the package raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CODAE_HIP_LIB: load another build of the library (same-box A/B of kernel variants)
LIB_PATH = os.environ.get("CODAE_HIP_LIB") or os.path.join(_HERE, "libcodae_hip.so")

PREC_F32 = 0
PREC_BF16 = 1

ABI_VERSION = 5
# CODAE_S_* of include/codae_hip.h (tests/test_host_logic.py parses the header and compares)
S_SQ_FULL, S_SQ_PARTIAL, S_GRAD_SQ, S_LAST_LOSS, S_STEP_SQ, S_CLIP_COEF = 0, 1, 2, 3, 4, 5
S_GRAD_SQ_SLOTS, S_N_SLOTS, S_ADAM_STEP, S_COUNT = 8, 64, 72, 80
KERNEL_CLASSES = ("gemm_fwd", "gemm_dgrad", "gemm_wgrad", "loss", "gather", "sumsq", "adam", "slab_reduce", "chain",
                  "bias_finish")


class HipError(RuntimeError):
    pass


class Spec(C.Structure):
    _fields_ = [("n_layers", C.c_int32),
                ("in_features", C.POINTER(C.c_int32)),
                ("out_features", C.POINTER(C.c_int32)),
                ("relu", C.POINTER(C.c_uint8)),
                ("max_batch", C.c_int32),
                ("precision", C.c_int32)]


class Sizes(C.Structure):
    _fields_ = [("n_param", C.c_int64), ("n_weight", C.c_int64), ("act_bytes", C.c_int64),
                ("dact_bytes", C.c_int64), ("slab_bytes", C.c_int64), ("bias_part_bytes", C.c_int64),
                ("n_scalars", C.c_int32)]


class Buffers(C.Structure):
    _fields_ = [("params", C.c_void_p), ("grads", C.c_void_p), ("adam_m", C.c_void_p),
                ("adam_v", C.c_void_p), ("shadow_w", C.c_void_p), ("acts", C.c_void_p),
                ("dacts", C.c_void_p), ("slabs", C.c_void_p), ("scalars", C.c_void_p), ("shadow_wt", C.c_void_p),
                ("bias_parts", C.c_void_p)]


class Batch(C.Structure):
    _fields_ = [("data", C.c_void_p), ("row_idx", C.c_void_p), ("mask_id", C.c_void_p),
                ("mask_table", C.c_void_p), ("B", C.c_int32), ("io", C.c_int32),
                ("mask_to_use", C.c_void_p), ("nb_run", C.c_int32), ("run", C.c_int32)]


class Hyper(C.Structure):
    _fields_ = [("lr", C.c_float), ("weight_decay", C.c_float), ("beta1", C.c_float),
                ("beta2", C.c_float), ("eps", C.c_float), ("max_grad_norm", C.c_float),
                ("step", C.c_int32), ("loss_scale_rows", C.c_float)]


_P, _I32, _I64, _F = C.c_void_p, C.c_int32, C.c_int64, C.c_float

# name -> (restype, argtypes); mirrors include/codae_hip.h one to one
PROTOTYPES = {
    "codae_last_error": (C.c_char_p, []),
    "codae_abi_version": (C.c_int, []),
    "codae_struct_sizes": (C.c_int, [C.POINTER(C.c_int32), _I32]),
    "codae_reload_env": (C.c_int, []),
    "codae_create": (C.c_int, [C.POINTER(Spec), C.POINTER(_P)]),
    "codae_destroy": (C.c_int, [_P]),
    "codae_get_sizes": (C.c_int, [_P, C.POINTER(Sizes)]),
    "codae_param_offsets": (C.c_int, [_P, _I32, C.POINTER(_I64), C.POINTER(_I64), C.POINTER(_I64)]),
    "codae_forward": (C.c_int, [_P, C.POINTER(Buffers), _P, _P, _I32, _I32, _I32, _I32, _P]),
    "codae_backward": (C.c_int, [_P, C.POINTER(Buffers), _P, _P, _I32, _I32, _I32, _P]),
    "codae_sync_shadows": (C.c_int, [_P, C.POINTER(Buffers), _P]),
    "codae_step_forward_loss": (C.c_int, [_P, C.POINTER(Buffers), C.POINTER(Batch), C.POINTER(Hyper), _P, _P]),
    "codae_step_backward": (C.c_int, [_P, C.POINTER(Buffers), _I32, _I32, _I32, _P]),
    "codae_step_update": (C.c_int, [_P, C.POINTER(Buffers), C.POINTER(Hyper), _P]),
    "codae_step_backward_async": (C.c_int, [_P, C.POINTER(Buffers), _I32, _I32, _I32, _P]),
    "codae_side_stream": (C.c_int, [_P, C.POINTER(C.c_void_p)]),
    "codae_profile_stride": (C.c_int, [_P, _I32]),
    "codae_join": (C.c_int, [_P, _P]),
    "codae_span_sumsq": (C.c_int, [_P, _I64, _P, _P]),
    "codae_step_update_span": (C.c_int, [_P, C.POINTER(Buffers), C.POINTER(Hyper), _I64, _I64, _P, _P]),
    "codae_sync_transposed": (C.c_int, [_P, C.POINTER(Buffers), _P]),
    "codae_train_step": (C.c_int, [_P, C.POINTER(Buffers), C.POINTER(Batch), C.POINTER(Hyper), _P]),
    "codae_train_step_graph": (C.c_int, [_P, C.POINTER(Buffers), C.POINTER(Batch), C.POINTER(Hyper), _P]),
    "codae_eval_step": (C.c_int, [_P, C.POINTER(Buffers), C.POINTER(Batch), _P, _P]),
    "codae_profile_begin": (C.c_int, [_P, C.c_uint32, _I32]),
    "codae_profile_end": (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_float), _I32, C.POINTER(C.c_int32)]),
    "codae_corrupt": (C.c_int, [_P, _P, _P, _I64, _P]),
    "codae_expand_masks": (C.c_int, [_P, _P, _P, _I32, _I32, _I32, _P, _P, _P]),
    "codae_mse_loss_fwd_bwd": (C.c_int, [_P, _P, _P, _P, _I64, _F, _P, _P]),
    "codae_clip_adam": (C.c_int, [_P, _P, _P, _P, _I64, C.POINTER(Hyper), _P, _P]),
    "codae_combined_loss_fwd_bwd": (C.c_int, [_P, _P, _I32, _I32, _I32, _P, _P, _P, _P, _P, _P, _P, _P]),
    "codae_combined_loss_full": (C.c_int, [_P, _P, _I32, _I32, _I32, _P, _P, _P, _P, _P]),
    "codae_row_norms": (C.c_int, [_P, _I64, _I32, _P, _P]),
    "codae_ranking_loss": (C.c_int, [_P, _P, _P, _I32, _I32, _I32, _I32, _P, _P, _I64, _P, _I32, _P, _P]),
    "codae_ranking_loss_batched": (C.c_int, [_P, _I32, _I32, _I32, _I32, _P, _P, _P, _I32, _I32, _P, _P, _P, _I64, _P, _P, _P, _P, _I32, _P,
                                             _I32, _P, _P, _P, _P, _P]),
    "codae_dp_unique_id": (C.c_int, [_P, _I32]),
    "codae_dp_init": (C.c_int, [_P, _P, _I32, _I32]),
    "codae_dp_destroy": (C.c_int, [_P]),
    "codae_train_step_dp": (C.c_int, [_P, C.POINTER(Buffers), C.POINTER(Batch), C.POINTER(Hyper), _I32, _P, _P, _P]),
    "codae_monitor_accumulate": (C.c_int, [_P, _P, _I32, _I32, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _I32, _P, _P]),
    "codae_gather_inventory_rows": (C.c_int, [_P, _I64, _I32, _I32, _P, _I32, _P, _P]),
    "codae_step_path": (C.c_int, [_P, _P, _I32]),
    "codae_linear_f32": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _I32, _I32, _P]),
    "codae_dgrad_f32": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _I32, _P]),
    "codae_wgrad_f32": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _I32, _P]),
    "codae_linear_bf16": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _P]),
    "codae_dgrad_bf16": (C.c_int, [_P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _P]),
    "codae_wgrad_bf16": (C.c_int, [_P, _P, _P, _P, _I64, _I32, _I32, _I32, _P]),
    "codae_cast_f32_to_bf16": (C.c_int, [_P, _P, _I64, _P]),
    "codae_debug_gemm_timeline": (C.c_int, [_P, _I32]),
    "codae_transpose_bf16": (C.c_int, [_P, _P, _I32, _I32, _P]),
}

_lib = None


def lib():
    """Load libcodae_hip.so (once).  Raises HipError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipError(
            "libcodae_hip.so is missing (%s). Build it with `python -c \"import __graft_entry__ as g; "
            "g.build()\"` or `make -C mui-deepautoencoder_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    import torch  # noqa: F401  -- loads the HIP runtime this library must share with torch
    handle = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(handle, name)
        fn.restype = res
        fn.argtypes = args
    if handle.codae_abi_version() != ABI_VERSION:
        raise HipError("libcodae_hip.so ABI version %d, this binding is written for %d: rebuild the library"
                       % (handle.codae_abi_version(), ABI_VERSION))
    check_struct_sizes(handle)
    _lib = handle
    return _lib


def struct_sizes_expected():
    """What this binding declares, in codae_struct_sizes() order."""
    return [C.sizeof(Spec), C.sizeof(Sizes), C.sizeof(Buffers), C.sizeof(Batch), C.sizeof(Hyper), S_COUNT,
            len(KERNEL_CLASSES)]


def check_struct_sizes(handle):
    """A ctypes struct shorter than the library's makes the engine read past the caller's memory
    (codae_buffers gained shadow_wt in ABI 2): compare every layout before the first call."""
    n = 7
    got = (C.c_int32 * n)()
    rc = handle.codae_struct_sizes(got, n)
    if rc != 0:
        raise HipError("codae_struct_sizes failed (%d)" % rc)
    want = struct_sizes_expected()
    names = ("codae_spec", "codae_sizes", "codae_buffers", "codae_batch", "codae_hyper", "CODAE_S_COUNT", "CODAE_K_COUNT")
    bad = ["%s: library %d, binding %d" % (nm, g, w) for nm, g, w in zip(names, list(got), want) if g != w]
    if bad:
        raise HipError("libcodae_hip.so and its ctypes binding disagree: " + "; ".join(bad))


def check(rc):
    if rc != 0:
        msg = lib().codae_last_error()
        raise HipError("codae_hip error %d: %s" % (rc, msg.decode() if msg else "?"))


def ptr(t):
    """data pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def current_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
