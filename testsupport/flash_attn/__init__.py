"""Comparator with the Python API of Dao-AILab `flash_attn` (>= 2.1 semantics), evaluated with plain torch math.

TEST / BENCHMARK INFRASTRUCTURE ONLY -- nothing under mini-flash-attention_amd/ imports this package.

Why it exists: the reference's benchmark scripts and several of its tests
(`benchmark/*.py`, `tests/test_varlen.py`, `tests/test_flash_decoding.py`, `tests/test_minimal.py`,
`tests/test_both_seqlens.py`, `tests/test_output_compare.py`) compare `mini_flash_attention` with the official
`flash_attn` wheel, which cannot be installed here (no network, CUDA-only).  Putting `testsupport/` on
PYTHONPATH makes `from flash_attn import flash_attn_func, flash_attn_varlen_func, flash_attn_with_kvcache` resolve
to this module, so those callers run unmodified on ROCm with a numerically trustworthy counterpart:

  * values come from fp32 softmax(Q K^T * scale + mask) V computed by torch (matmul / softmax, chunked over the
    batch and the query rows so the fp32 score block stays bounded), cast to q's dtype at the end;
  * masks follow flash-attn >= 2.1: causal and sliding windows are aligned to the BOTTOM-RIGHT corner
    (query row i of Sq sees keys <= i + Sk - Sq); rows that see no key give 0 and LSE = -inf;
  * GQA/MQA by head index (query head h reads KV head h // (Hq // Hkv));
  * the returned LSE is the natural-log LSE of the scaled scores, laid out like flash-attn 2.6+:
    (B, H, Sq) dense / kv-cache, (H, total_q) varlen;
  * dropout, alibi, softcap, rotary and fp8 descaling are NOT modelled: passing them raises NotImplementedError.

For plain dense attention on a GPU where the fp32 path would be slow (the benchmark shapes), `fast=True`
(or env FLASH_ATTN_SHIM_FAST=1) dispatches to torch's fused scaled_dot_product_attention in the input dtype when
the mask allows it -- that is the "vendor library" column of the benchmark, not a numerics reference.
"""
import math
import os

import torch
import torch.nn.functional as F

__version__ = "2.6.3+torchmath.shim"

_ROW_CHUNK_ELEMS = 1 << 26  # fp32 score elements per chunk (256 MiB)


def _unsupported(**kw):
    """kw: name=(value, default).  Anything other than None / the default is a feature this comparator lacks."""
    for name, (val, default) in kw.items():
        if val is None:
            continue
        if isinstance(val, (int, float)) and default is not None and float(val) == float(default):
            continue
        raise NotImplementedError(f"flash_attn comparator: `{name}` is not modelled (got {val!r})")


def _masked_attention(q, k, v, scale, causal, window, sk_valid=None):
    """q (Sq,Hq,D), k/v (Sk,Hk,D) of ONE sequence -> (out (Sq,Hq,D) fp32, lse (Hq,Sq) fp32).  Keys >= sk_valid
    are invisible.  Bottom-right alignment."""
    sq, hq, d = q.shape
    sk = k.shape[0] if sk_valid is None else int(sk_valid)
    hk = k.shape[1]
    g = hq // hk
    out = torch.zeros(sq, hq, d, dtype=torch.float32, device=q.device)
    lse = torch.full((hq, sq), float("-inf"), dtype=torch.float32, device=q.device)
    if sq == 0:
        return out, lse
    if sk <= 0:
        return out, lse
    kf = k[:sk].float().permute(1, 0, 2)  # (Hk, Sk, D)
    vf = v[:sk].float().permute(1, 0, 2)
    wl, wr = window
    if causal:
        wr = 0 if wr < 0 else min(wr, 0)
    rows_per_chunk = max(1, _ROW_CHUNK_ELEMS // max(1, hq * sk))
    kidx = torch.arange(sk, device=q.device)
    for r0 in range(0, sq, rows_per_chunk):
        r1 = min(sq, r0 + rows_per_chunk)
        qf = q[r0:r1].float().permute(1, 0, 2).reshape(hk, g, r1 - r0, d)  # (Hk, G, R, D)
        s = torch.einsum("hgrd,hkd->hgrk", qf, kf) * scale
        if wl >= 0 or wr >= 0:
            pos = torch.arange(r0, r1, device=q.device)[:, None] + (sk - sq)  # diagonal key of each row
            keep = torch.ones(r1 - r0, sk, dtype=torch.bool, device=q.device)
            if wr >= 0:
                keep &= kidx[None, :] <= pos + wr
            if wl >= 0:
                keep &= kidx[None, :] >= pos - wl
            s = s.masked_fill(~keep, float("-inf"))
        m = s.amax(dim=-1, keepdim=True)
        m = torch.where(torch.isinf(m), torch.zeros_like(m), m)
        p = torch.exp(s - m)
        l = p.sum(dim=-1, keepdim=True)
        o = torch.einsum("hgrk,hkd->hgrd", p, vf) / torch.where(l == 0, torch.ones_like(l), l)
        out[r0:r1] = o.reshape(hq, r1 - r0, d).permute(1, 0, 2)
        lse[:, r0:r1] = torch.where(l > 0, m + torch.log(l), torch.full_like(l, float("-inf"))).reshape(hq, r1 - r0)
    return out, lse


def _scale(q, softmax_scale):
    return float(softmax_scale) if softmax_scale is not None else 1.0 / math.sqrt(q.shape[-1])


def _fast_enabled(fast):
    return bool(fast) or os.environ.get("FLASH_ATTN_SHIM_FAST", "0") == "1"


def flash_attn_func(q, k, v, dropout_p=0.0, softmax_scale=None, causal=False, window_size=(-1, -1), softcap=0.0,
                    alibi_slopes=None, deterministic=False, return_attn_probs=False, *, fast=False):
    """q (B,Sq,Hq,D), k/v (B,Sk,Hk,D) -> out (B,Sq,Hq,D) [, lse (B,Hq,Sq), None]."""
    _unsupported(dropout_p=(dropout_p, 0.0), softcap=(softcap, 0.0), alibi_slopes=(alibi_slopes, None))
    scale = _scale(q, softmax_scale)
    B, Sq, Hq, D = q.shape
    Sk = k.shape[1]
    window = tuple(int(w) for w in window_size)
    if _fast_enabled(fast) and not return_attn_probs and window == (-1, -1) and (not causal or Sq == Sk):
        qt, kt, vt = (t.transpose(1, 2) for t in (q, k, v))
        o = F.scaled_dot_product_attention(qt, kt, vt, is_causal=causal, scale=scale, enable_gqa=Hq != k.shape[2])
        return o.transpose(1, 2).contiguous()
    out = torch.empty(B, Sq, Hq, D, dtype=q.dtype, device=q.device)
    lse = torch.empty(B, Hq, Sq, dtype=torch.float32, device=q.device)
    for b in range(B):
        o, l = _masked_attention(q[b], k[b], v[b], scale, causal, window)
        out[b] = o.to(q.dtype)
        lse[b] = l
    return (out, lse, None) if return_attn_probs else out


def _gather_pages(cache, table_row, length):
    """(num_blocks, page, Hk, D) + one row of the block table -> the first `length` keys, (length, Hk, D)."""
    page = cache.shape[1]
    nblk = (length + page - 1) // page
    if nblk == 0:
        return cache.new_zeros(0, cache.shape[2], cache.shape[3])
    return cache[table_row[:nblk].long()].reshape(nblk * page, cache.shape[2], cache.shape[3])[:length]


def flash_attn_varlen_func(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, dropout_p=0.0,
                           softmax_scale=None, causal=False, window_size=(-1, -1), softcap=0.0, alibi_slopes=None,
                           deterministic=False, return_attn_probs=False, block_table=None):
    """Packed q (total_q,Hq,D); k/v packed (total_k,Hk,D) or, with block_table (B,max_blocks), paged
    (num_blocks,page,Hk,D) -> out (total_q,Hq,D) [, lse (Hq,total_q), None]."""
    _unsupported(dropout_p=(dropout_p, 0.0), softcap=(softcap, 0.0), alibi_slopes=(alibi_slopes, None))
    scale = _scale(q, softmax_scale)
    window = tuple(int(w) for w in window_size)
    cq = cu_seqlens_q.tolist()
    ck = cu_seqlens_k.tolist()
    out = torch.zeros_like(q)
    lse = torch.full((q.shape[1], q.shape[0]), float("-inf"), dtype=torch.float32, device=q.device)
    for b in range(len(cq) - 1):
        q0, q1 = cq[b], cq[b + 1]
        lk = ck[b + 1] - ck[b]
        if q1 == q0:
            continue
        if block_table is not None:
            kb, vb = _gather_pages(k, block_table[b], lk), _gather_pages(v, block_table[b], lk)
        else:
            kb, vb = k[ck[b]:ck[b + 1]], v[ck[b]:ck[b + 1]]
        o, l = _masked_attention(q[q0:q1], kb, vb, scale, causal, window)
        out[q0:q1] = o.to(q.dtype)
        lse[:, q0:q1] = l
    return (out, lse, None) if return_attn_probs else out


def flash_attn_with_kvcache(q, k_cache, v_cache, k=None, v=None, rotary_cos=None, rotary_sin=None,
                            cache_seqlens=None, cache_batch_idx=None, cache_leftpad=None, block_table=None,
                            softmax_scale=None, causal=False, window_size=(-1, -1), softcap=0.0,
                            rotary_interleaved=True, alibi_slopes=None, num_splits=0, return_softmax_lse=False):
    """q (B,Sq,Hq,D); cache dense (B,Sk,Hk,D) or paged (num_blocks,page,Hk,D) + block_table.  With k/v
    (B,Sn,Hk,D) the new rows are first written IN PLACE at cache_seqlens[b] (as flash-attn does) and attended to.
    -> out (B,Sq,Hq,D) [, lse (B,Hq,Sq)]."""
    _unsupported(rotary_cos=(rotary_cos, None), rotary_sin=(rotary_sin, None), cache_batch_idx=(cache_batch_idx, None),
                 cache_leftpad=(cache_leftpad, None), softcap=(softcap, 0.0), alibi_slopes=(alibi_slopes, None))
    scale = _scale(q, softmax_scale)
    window = tuple(int(w) for w in window_size)
    B, Sq, Hq, D = q.shape
    if cache_seqlens is None:
        cap = k_cache.shape[1] if block_table is None else block_table.shape[1] * k_cache.shape[1]
        lens = [cap] * B
    elif isinstance(cache_seqlens, int):
        lens = [cache_seqlens] * B
    else:
        lens = cache_seqlens.tolist()
    if k is not None:
        sn = k.shape[1]
        page = k_cache.shape[1]
        for b in range(B):
            if block_table is None:
                n = max(0, min(sn, k_cache.shape[1] - lens[b]))
                k_cache[b, lens[b]:lens[b] + n], v_cache[b, lens[b]:lens[b] + n] = k[b, :n], v[b, :n]
            else:
                for t in range(sn):
                    pos = lens[b] + t
                    if pos // page < block_table.shape[1]:
                        blk = int(block_table[b, pos // page])
                        k_cache[blk, pos % page], v_cache[blk, pos % page] = k[b, t], v[b, t]
            lens[b] += sn
    if (_fast_enabled(False) and not return_softmax_lse and block_table is None and window == (-1, -1)
            and (Sq == 1 or not causal) and len(set(lens)) == 1 and 0 < lens[0] <= k_cache.shape[1]):
        # benchmark column: torch's fused attention over the valid prefix of a uniform dense cache
        n = lens[0]
        o = F.scaled_dot_product_attention(q.transpose(1, 2), k_cache[:, :n].transpose(1, 2), v_cache[:, :n].transpose(1, 2),
                                           scale=scale, enable_gqa=Hq != k_cache.shape[2])
        return o.transpose(1, 2).contiguous()
    out = torch.empty_like(q)
    lse = torch.empty(B, Hq, Sq, dtype=torch.float32, device=q.device)
    for b in range(B):
        if block_table is None:
            kb, vb = k_cache[b], v_cache[b]
            valid = min(lens[b], k_cache.shape[1])
        else:
            valid = min(lens[b], block_table.shape[1] * k_cache.shape[1])
            kb, vb = _gather_pages(k_cache, block_table[b], valid), _gather_pages(v_cache, block_table[b], valid)
        o, l = _masked_attention(q[b], kb, vb, scale, causal, window, sk_valid=valid)
        out[b] = o.to(q.dtype)
        lse[b] = l
    return (out, lse) if return_softmax_lse else out
