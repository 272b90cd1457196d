"""ctypes binding of the C ABI in include/mfa.h (libmfa_hip.so) — for hosts that do not go through the
torch extension (another framework's tensors, raw hipMalloc pointers) and for the parity tests, which call
the library exactly as a foreign host would.

`import torch` BEFORE loading when both live in one process: torch bundles its own HIP runtime and the
loader then shares it with libmfa_hip.so (same SONAME); see INTEGRATION.md.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmfa_hip.so")

MFA_OK = 0
MFA_ERR_INVALID_ARGUMENT = -1
MFA_ERR_UNSUPPORTED = -2
MFA_ERR_LAUNCH = -3
MFA_ERR_WORKSPACE = -4
MFA_SPLIT_COUNTERS_MAX = 65536
MFA_ROUTE_DECODE, MFA_ROUTE_PACKED, MFA_ROUTE_PREFILL, MFA_ROUTE_COMBINE_LAUNCH, MFA_ROUTE_FUSED_MERGE, MFA_ROUTE_PREFILL64 = 1, 2, 4, 8, 16, 32

_i64, _i32, _f32, _vp = ctypes.c_int64, ctypes.c_int32, ctypes.c_float, ctypes.c_void_p


class ForwardParams(ctypes.Structure):
    """struct mfa_forward_params (include/mfa.h), field for field."""

    _fields_ = [
        ("q_ptr", _vp), ("k_ptr", _vp), ("v_ptr", _vp), ("o_ptr", _vp),
        ("q_batch_stride", _i64), ("q_head_stride", _i64), ("q_row_stride", _i64),
        ("k_batch_stride", _i64), ("k_head_stride", _i64), ("k_row_stride", _i64),
        ("v_batch_stride", _i64), ("v_head_stride", _i64), ("v_row_stride", _i64),
        ("o_batch_stride", _i64), ("o_head_stride", _i64), ("o_row_stride", _i64),
        ("is_causal", _i32), ("window_size_left", _i32), ("window_size_right", _i32),
        ("heads", _i32), ("kv_heads", _i32), ("kv_group_size", _i32),
        ("batch", _i32), ("seqlen_q", _i32), ("seqlen_k", _i32),
        ("head_dim", _i32), ("softmax_scale", _f32), ("softmax_scale_log2", _f32),
        ("is_bf16", _i32),
        ("cu_seqlens_q", _vp), ("cu_seqlens_k", _vp),
        ("block_table", _vp), ("block_table_batch_stride", _i64), ("page_block_size", _i32),
        ("k_cache_block_stride", _i64), ("v_cache_block_stride", _i64),
        ("seqlens_k", _vp),
        ("num_splits", _i32),
        ("softmax_lse_ptr", _vp), ("softmax_lseaccum_ptr", _vp), ("oaccum_ptr", _vp),
        ("max_blocks_per_seq", _i32), ("num_cus", _i32), ("mask_bottom_right", _i32),
        ("use_local_window", _i32), ("local_window_left", _i32), ("local_window_right", _i32),
        ("total_q", _i64), ("seqlens_k_offset", _i32), ("reserved", _i32),
        ("split_counters", _vp), ("split_counters_len", _i64),
    ]


class KvAppendParams(ctypes.Structure):
    """struct mfa_kvcache_append_params (include/mfa.h)."""

    _fields_ = [
        ("k_new", _vp), ("v_new", _vp), ("k_cache", _vp), ("v_cache", _vp),
        ("kn_batch_stride", _i64), ("kn_row_stride", _i64), ("kn_head_stride", _i64),
        ("vn_batch_stride", _i64), ("vn_row_stride", _i64), ("vn_head_stride", _i64),
        ("kc_batch_stride", _i64), ("kc_row_stride", _i64), ("kc_head_stride", _i64),
        ("vc_batch_stride", _i64), ("vc_row_stride", _i64), ("vc_head_stride", _i64),
        ("seqlens_k", _vp), ("block_table", _vp), ("block_table_batch_stride", _i64),
        ("batch", _i32), ("seqlen_new", _i32), ("kv_heads", _i32), ("head_dim", _i32),
        ("seqlen_k", _i32), ("page_block_size", _i32), ("max_blocks_per_seq", _i32), ("is_bf16", _i32),
    ]


_lib = None


def load(path: str = LIB_PATH) -> ctypes.CDLL:
    """Load libmfa_hip.so (once) and declare the prototypes of include/mfa.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise OSError(f"{path} is not built; run `python mini-flash-attention_amd/build.py` (no fallback exists)")
    lib = ctypes.CDLL(path)
    P = ctypes.POINTER(ForwardParams)
    lib.mfa_abi_version.restype = ctypes.c_int
    lib.mfa_version.restype = ctypes.c_char_p
    lib.mfa_last_error.restype = ctypes.c_char_p
    lib.mfa_forward_params_sizeof.restype = ctypes.c_size_t
    lib.mfa_forward_params_set_scale.argtypes = [P]
    lib.mfa_forward_params_set_scale.restype = None
    lib.mfa_run_flash_attention_forward.argtypes = [P, _vp]
    lib.mfa_run_flash_attention_forward.restype = ctypes.c_int
    lib.mfa_run_flash_attention_with_kv_cache.argtypes = [P, _vp]
    lib.mfa_run_flash_attention_with_kv_cache.restype = ctypes.c_int
    lib.mfa_kvcache_append.argtypes = [ctypes.POINTER(KvAppendParams), _vp]
    lib.mfa_kvcache_append.restype = ctypes.c_int
    lib.mfa_kvcache_append_params_sizeof.restype = ctypes.c_size_t
    lib.mfa_num_splits_heuristic.argtypes = [ctypes.c_int] * 5
    lib.mfa_num_splits_heuristic.restype = ctypes.c_int
    lib.mfa_decode_workspace_bytes.argtypes = [ctypes.c_int] * 4 + [ctypes.POINTER(ctypes.c_size_t)] * 2
    lib.mfa_decode_workspace_bytes.restype = None
    lib.mfa_kvcache_plan.argtypes = [P, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]
    lib.mfa_kvcache_plan.restype = ctypes.c_int
    lib.mfa_init.argtypes = [ctypes.c_int]
    lib.mfa_init.restype = ctypes.c_int
    lib.mfa_kvcache_counter_count.argtypes = [P]
    lib.mfa_kvcache_counter_count.restype = ctypes.c_size_t
    lib.mfa_debug_last_route.restype = ctypes.c_int
    lib.mfa_stream_is_capturing.argtypes = [_vp]
    lib.mfa_stream_is_capturing.restype = ctypes.c_int
    lib.mfa_device_cu_count.argtypes = [ctypes.c_int]
    lib.mfa_device_cu_count.restype = ctypes.c_int
    _lib = lib
    return lib


def last_error() -> str:
    return load().mfa_last_error().decode()
