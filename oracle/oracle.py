"""Python face of the CPU oracle.  TEST INFRASTRUCTURE ONLY (see the header of mfa_oracle.c).

Two independent checkers live here:
  * `restated_*`  — the C restatement of the reference's algorithm (oracle/mfa_oracle.c) through ctypes;
  * `sdpa_*`      — torch.nn.functional.scaled_dot_product_attention in fp32 on the CPU, i.e. the oracle the
                    reference's own tests use (reference tests/test_mha.py:75-91, test_gqa.py:118-128).
Nothing under mini-flash-attention_amd/ imports this module.
"""
import ctypes
import os
import subprocess
import sys

import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
_SRC = os.path.join(_HERE, "mfa_oracle.c")
_LIB = os.path.join(_HERE, "_build", "libmfa_oracle.so")

def _load_capi_mirror():
    """The ctypes mirror of struct mfa_forward_params lives in the product's capi.py (pure Python, loads no
    native code on import).  Load that one file without importing the package (which needs the built HIP
    extension) and share the module object with a later `import mini_flash_attention.capi`."""
    import importlib.util
    name = "mini_flash_attention.capi"
    if name in sys.modules:
        return sys.modules[name]
    path = os.path.join(_ROOT, "mini-flash-attention_amd", "mini_flash_attention", "capi.py")
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


_capi = _load_capi_mirror()
ForwardParams = _capi.ForwardParams


def build(force: bool = False) -> str:
    """gcc the restatement into oracle/_build/libmfa_oracle.so (plain C, OpenMP over (batch, head))."""
    if os.environ.get("MFA_TEST_ORACLE_LIB"):  # tests/test_sanitizers_cpu.py: a build with -fsanitize=address,undefined
        return os.environ["MFA_TEST_ORACLE_LIB"]
    hdr = os.path.join(_ROOT, "include", "mfa.h")
    if not force and os.path.exists(_LIB) and os.path.getmtime(_LIB) >= max(os.path.getmtime(_SRC), os.path.getmtime(hdr)):
        return _LIB
    os.makedirs(os.path.dirname(_LIB), exist_ok=True)
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-fopenmp", _SRC, "-o", _LIB, "-lm"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        P = ctypes.POINTER(ForwardParams)
        _lib.mfa_oracle_prefill.argtypes = [P]
        _lib.mfa_oracle_decode.argtypes = [P]
        _lib.mfa_oracle_f32_to_f16.argtypes = [ctypes.c_float]
        _lib.mfa_oracle_f32_to_f16.restype = ctypes.c_uint16
        _lib.mfa_oracle_f32_to_bf16.argtypes = [ctypes.c_float]
        _lib.mfa_oracle_f32_to_bf16.restype = ctypes.c_uint16
        _lib.mfa_oracle_f16_to_f32.argtypes = [ctypes.c_uint16]
        _lib.mfa_oracle_f16_to_f32.restype = ctypes.c_float
        _lib.mfa_oracle_bf16_to_f32.argtypes = [ctypes.c_uint16]
        _lib.mfa_oracle_bf16_to_f32.restype = ctypes.c_float
    return _lib


def fill_params(q, k, v, o, *, causal=False, cu_q=None, cu_k=None, max_sq=None, max_sk=None, block_table=None,
                seqlens_k=None, num_splits=1):
    """Fill a ForwardParams exactly as the reference's forward_params_init does (api.cpp:30-101) from torch
    tensors that share one device.  Works for CPU tensors (oracle) and GPU tensors (C ABI tests)."""
    p = ForwardParams()
    p.q_ptr, p.k_ptr, p.v_ptr, p.o_ptr = q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr()
    for name, t in (("q", q), ("k", k), ("v", v), ("o", o)):
        setattr(p, f"{name}_row_stride", t.stride(-3))
        setattr(p, f"{name}_head_stride", t.stride(-2))
        if t.dim() == 4 and cu_q is None and not (block_table is not None and name in "kv"):
            setattr(p, f"{name}_batch_stride", t.stride(0))
    p.is_bf16 = int(q.dtype == torch.bfloat16)
    p.is_causal = int(causal)
    p.window_size_left, p.window_size_right = -1, (0 if causal else -1)
    p.heads, p.kv_heads, p.head_dim = q.size(-2), k.size(-2), q.size(-1)
    if cu_q is not None:
        p.batch = cu_q.numel() - 1
        p.seqlen_q, p.seqlen_k = int(max_sq), int(max_sk)
        p.cu_seqlens_q, p.cu_seqlens_k = cu_q.data_ptr(), cu_k.data_ptr()
    else:
        p.batch, p.seqlen_q = q.size(0), q.size(1)
        p.seqlen_k = k.size(1) if block_table is None else block_table.size(1) * k.size(1)
    if block_table is not None:
        p.block_table = block_table.data_ptr()
        p.block_table_batch_stride = block_table.stride(0)
        p.max_blocks_per_seq = block_table.size(1)
        p.page_block_size = k.size(1)
        p.k_cache_block_stride, p.v_cache_block_stride = k.stride(0), v.stride(0)
    if seqlens_k is not None:
        p.seqlens_k = seqlens_k.data_ptr()
    p.num_splits = int(num_splits)
    p.kv_group_size = p.heads // p.kv_heads
    p.softmax_scale = 1.0 / (p.head_dim ** 0.5)
    p.softmax_scale_log2 = p.softmax_scale * 1.4426950408889634
    return p


# ---- the C restatement -------------------------------------------------------------------------
def restated_prefill(q, k, v, causal=False, cu_q=None, cu_k=None, max_sq=None, max_sk=None, block_table=None):
    q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
    o = torch.empty_like(q)
    p = fill_params(q, k, v, o, causal=causal, cu_q=cu_q, cu_k=cu_k, max_sq=max_sq, max_sk=max_sk,
                    block_table=block_table)
    rc = lib().mfa_oracle_prefill(ctypes.byref(p))
    assert rc == 0
    return o


def restated_decode(q, k_cache, v_cache, seqlens_k=None, block_table=None, num_splits=1, return_partials=False):
    q, k_cache, v_cache = q.contiguous(), k_cache.contiguous(), v_cache.contiguous()
    o = torch.empty_like(q)
    p = fill_params(q, k_cache, v_cache, o, seqlens_k=seqlens_k, block_table=block_table, num_splits=num_splits)
    B, H, D = q.size(0), q.size(2), q.size(3)
    lse = torch.empty(B, H, dtype=torch.float32)
    p.softmax_lse_ptr = lse.data_ptr()
    S = max(1, num_splits)
    o_acc = torch.zeros(S, B, H, D, dtype=torch.float32)
    lse_acc = torch.full((S, B, H), float("-inf"), dtype=torch.float32)
    if S > 1:
        p.oaccum_ptr, p.softmax_lseaccum_ptr = o_acc.data_ptr(), lse_acc.data_ptr()
    rc = lib().mfa_oracle_decode(ctypes.byref(p))
    assert rc == 0
    return (o, lse, o_acc, lse_acc) if return_partials else o


# ---- torch SDPA, fp32, CPU: the reference tests' own oracle ---------------------------------------
def sdpa_dense(q, k, v, causal=False):
    """q (B,Sq,H,D), k/v (B,Sk,Hk,D) -> (B,Sq,H,D) fp32.  GQA by repeat_interleave (tests/test_gqa.py:118-120);
    is_causal=True is top-left aligned like the reference kernel (prefill.cuh:416-419)."""
    g = q.size(2) // k.size(2)
    qf, kf, vf = (t.float().transpose(1, 2) for t in (q, k, v))
    if g > 1:
        kf, vf = kf.repeat_interleave(g, dim=1), vf.repeat_interleave(g, dim=1)
    if k.size(1) == 0:
        return torch.zeros(q.shape, dtype=torch.float32)
    return F.scaled_dot_product_attention(qf, kf, vf, is_causal=causal).transpose(1, 2).contiguous()


def gather_pages(cache, block_table, b, length):
    """Logical rows [0, length) of sequence b from a paged cache (num_blocks, page, Hk, D)."""
    page = cache.size(1)
    nblk = (length + page - 1) // page
    rows = cache[block_table[b, :nblk].long()].reshape(nblk * page, cache.size(2), cache.size(3))
    return rows[:length]


def sdpa_varlen(q, k, v, cu_q, cu_k, causal=False, block_table=None):
    outs = []
    for b in range(cu_q.numel() - 1):
        q0, q1, k0, k1 = int(cu_q[b]), int(cu_q[b + 1]), int(cu_k[b]), int(cu_k[b + 1])
        kb = gather_pages(k, block_table, b, k1 - k0) if block_table is not None else k[k0:k1]
        vb = gather_pages(v, block_table, b, k1 - k0) if block_table is not None else v[k0:k1]
        outs.append(sdpa_dense(q[q0:q1][None], kb[None], vb[None], causal)[0])
    return torch.cat(outs) if outs else torch.zeros(q.shape, dtype=torch.float32)


def sdpa_decode(q, k_cache, v_cache, seqlens_k=None, block_table=None):
    B = q.size(0)
    outs = []
    for b in range(B):
        if block_table is not None:
            full = block_table.size(1) * k_cache.size(1)
            n = int(seqlens_k[b]) if seqlens_k is not None else full
            kb, vb = gather_pages(k_cache, block_table, b, n), gather_pages(v_cache, block_table, b, n)
        else:
            n = int(seqlens_k[b]) if seqlens_k is not None else k_cache.size(1)
            kb, vb = k_cache[b, :n], v_cache[b, :n]
        outs.append(sdpa_dense(q[b:b + 1], kb[None], vb[None], False)[0])
    return torch.stack(outs)
