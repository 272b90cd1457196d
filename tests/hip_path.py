"""Two ways into the HIP path for the GPU parity tests:
  route "api"  : mini_flash_attention.flash_attn_* (Python API -> _C torch extension -> C ABI)
  route "capi" : ctypes straight into libmfa_hip.so with raw device pointers and the raw HIP stream, exactly
                 as a non-torch host would call include/mfa.h (torch is only the allocator here).
"""
import ctypes

import torch
import torch.nn.functional as F

from oracle.oracle import fill_params

ROUTES = ("api", "capi")


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


_COUNTERS = {}


def split_counters(capi, p):
    """What a non-torch host does for the in-kernel split merge (include/mfa.h): mfa_init() once per device, one zeroed
    int32 buffer per stream that lives as long as the process, handed over when the plan wants counters."""
    lib = capi.load()
    if lib.mfa_kvcache_counter_count(ctypes.byref(p)) == 0 or lib.mfa_init(-1) != 1:
        return
    key = (torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream)
    if key not in _COUNTERS:
        _COUNTERS[key] = torch.zeros(capi.MFA_SPLIT_COUNTERS_MAX, dtype=torch.int32, device="cuda")
    p.split_counters, p.split_counters_len = _COUNTERS[key].data_ptr(), capi.MFA_SPLIT_COUNTERS_MAX


def _check(capi, rc):
    assert rc == 0, f"C ABI returned {rc}: {capi.last_error()}"


def prefill(route, mfa, capi, q, k, v, causal=False, cu_q=None, cu_k=None, max_sq=None, max_sk=None, block_table=None):
    if route == "api":
        if cu_q is None:
            return mfa.flash_attn_func(q, k, v, causal=causal)
        return mfa.flash_attn_varlen_func(q, k, v, cu_q, cu_k, max_sq, max_sk, causal=causal, block_table=block_table)
    o = torch.empty_like(q)
    p = fill_params(q, k, v, o, causal=causal, cu_q=cu_q, cu_k=cu_k, max_sq=max_sq, max_sk=max_sk, block_table=block_table)
    if cu_q is not None:
        p.total_q = q.size(0)  # (as the torch binding does: lets the launcher see how even the batch is)
    _check(capi, capi.load().mfa_run_flash_attention_forward(ctypes.byref(p), _stream()))
    return o


def decode(route, mfa, capi, q, kc, vc, lens=None, block_table=None, num_splits=0, return_partials=False, counters=True, ws=None):
    """counters=False: no arrival counters are handed over (the merge is decode_combine_kernel's launch); ws: (o_acc, lse_acc)
    workspaces to reuse instead of fresh ones (the same addresses launch after launch)"""
    if route == "api" and not return_partials:
        return mfa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, block_table=block_table, num_splits=num_splits)
    lib = capi.load()
    o = torch.empty_like(q)
    p = fill_params(q, kc, vc, o, seqlens_k=lens, block_table=block_table)
    B, H, D = q.size(0), q.size(2), q.size(3)
    p.num_splits = lib.mfa_num_splits_heuristic(int(num_splits), B, kc.size(-2), p.seqlen_k, 0)
    lse = torch.empty(B, H, dtype=torch.float32, device=q.device)
    p.softmax_lse_ptr = lse.data_ptr()
    S = p.num_splits
    if ws is not None:
        o_acc, lse_acc = ws
        assert o_acc.numel() >= max(S, 1) * B * H * D and lse_acc.numel() >= max(S, 1) * B * H
    else:
        o_acc = torch.empty(max(S, 1), B, H, D, dtype=torch.float32, device=q.device)
        lse_acc = torch.empty(max(S, 1), B, H, dtype=torch.float32, device=q.device)
    if S > 1:
        p.oaccum_ptr, p.softmax_lseaccum_ptr = o_acc.data_ptr(), lse_acc.data_ptr()
        if counters:
            split_counters(capi, p)
    _check(capi, lib.mfa_run_flash_attention_with_kv_cache(ctypes.byref(p), _stream()))
    return (o, lse, o_acc, lse_acc, S) if return_partials else o


def sdpa_gpu(q, k, v, causal=False):
    """torch SDPA evaluated in fp32 on the GPU (math identical to oracle.sdpa_dense), for sizes where the CPU
    oracle would take too long; GQA via repeat_interleave as reference tests/test_gqa.py:118-120."""
    g = q.size(2) // k.size(2)
    qf, kf, vf = (t.float().transpose(1, 2) for t in (q, k, v))
    if g > 1:
        kf, vf = kf.repeat_interleave(g, dim=1), vf.repeat_interleave(g, dim=1)
    from torch.nn.attention import SDPBackend, sdpa_kernel
    with sdpa_kernel(SDPBackend.MATH):
        return F.scaled_dot_product_attention(qf, kf, vf, is_causal=causal).transpose(1, 2)


def make_paged(kc, vc, page, seed=0, extra_blocks=2):
    """Scatter dense (B,Sk,Hk,D) caches into a permuted page pool; returns (k_pages, v_pages, block_table)."""
    B, Sk, Hk, D = kc.shape
    nb = (Sk + page - 1) // page
    g = torch.Generator(device="cpu").manual_seed(seed)
    perm = torch.randperm(B * nb + extra_blocks, generator=g)[: B * nb].to(kc.device)
    pad = nb * page - Sk
    kp = torch.randn(B * nb + extra_blocks, page, Hk, D, device=kc.device, dtype=torch.float32).to(kc.dtype)
    vp = torch.randn(B * nb + extra_blocks, page, Hk, D, device=kc.device, dtype=torch.float32).to(kc.dtype)
    kp[perm] = F.pad(kc, (0, 0, 0, 0, 0, pad)).reshape(B * nb, page, Hk, D)
    vp[perm] = F.pad(vc, (0, 0, 0, 0, 0, pad)).reshape(B * nb, page, Hk, D)
    return kp, vp, perm.int().view(B, nb).contiguous()
