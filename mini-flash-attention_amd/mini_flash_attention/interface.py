"""Python API: three thin, positional forwards to ``mini_flash_attention._C``.

Mirrors reference mini_flash_attention/interface.py:6-124 (names, argument order, defaults, return
value).  Differences are supersets only:
  * ``flash_attn_with_kvcache`` accepts ``causal=`` (the reference's own decode tests pass it,
    tests/test_flash_decoding.py:70); with seqlen_q == 1 the single query is the last position, so the
    flag cannot change the result and is ignored, exactly as the reference C++ does (api.cpp:349);
  * ``cache_seqlens`` may be ``None`` (= full cache) or an ``int`` (broadcast), where the reference
    crashes or raises.
"""
from typing import Optional, Union

import torch

try:
    import mini_flash_attention._C as _C  # type: ignore[import-not-found]
except ImportError as e:  # fail loudly: there is no fallback path
    raise ImportError(
        "mini_flash_attention._C is not built. Run `python mini-flash-attention_amd/build.py` "
        "(hipcc, --offload-arch=gfx950) first; there is no CPU/eager fallback."
    ) from e


def _window(window_size):
    left, right = window_size
    return int(left), int(right)


def flash_attn_func(
    q: torch.Tensor,
    k: torch.Tensor,
    v: torch.Tensor,
    causal: bool = False,
    window_size=(-1, -1),
    return_softmax_lse: bool = False,
):
    """Dense attention forward, O = softmax(Q K^T / sqrt(D) + mask) V.

    q: (batch, seqlen_q, nheads, headdim); k, v: (batch, seqlen_k, nheads_k, headdim), fp16 or bf16 on
    the GPU, nheads % nheads_k == 0 (MQA/GQA: query head h uses KV head h // (nheads // nheads_k)).
    causal: top-left aligned mask (key index > query index is masked), as the reference kernel and
    torch SDPA ``is_causal=True``.
    window_size (superset): (left, right) sliding window, query i sees keys [i - left, i + right]; -1 = unbounded.
    return_softmax_lse (superset): also return the natural-log LSE, (batch, nheads, seqlen_q) fp32.
    Returns (batch, seqlen_q, nheads, headdim) in q's dtype (and the LSE when asked).
    """
    if window_size == (-1, -1) and not return_softmax_lse:
        return _C.mini_flash_attention_forward(q, k, v, None, causal, -1, -1)
    left, right = _window(window_size)
    out, lse = _C.forward_ex(q, k, v, None, causal, left, right, False, return_softmax_lse)
    return (out, lse) if return_softmax_lse else out


def flash_attn_varlen_func(
    q: torch.Tensor,
    k: torch.Tensor,
    v: torch.Tensor,
    cu_seqlens_q: torch.Tensor,
    cu_seqlens_k: torch.Tensor,
    max_seqlen_q: int,
    max_seqlen_k: int,
    causal: bool = False,
    block_table=None,
    window_size=(-1, -1),
    return_softmax_lse: bool = False,
):
    """Packed variable-length attention forward (continuous batching).

    q: (total_q, nheads, headdim); k, v: (total_k, nheads_k, headdim), or with ``block_table``
    (batch, max_blocks) int32: paged (num_blocks, page_block_size, nheads_k, headdim).
    cu_seqlens_q / cu_seqlens_k: (batch + 1,) int32 cumulative lengths.
    window_size / return_softmax_lse: supersets, as in flash_attn_func (LSE layout (nheads, total_q)).
    Returns (total_q, nheads, headdim).
    """
    if window_size == (-1, -1) and not return_softmax_lse:
        return _C.mini_flash_attention_varlen_forward(
            q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, causal, -1, -1, block_table
        )
    left, right = _window(window_size)
    out, lse = _C.varlen_forward_ex(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, causal,
                                    left, right, False, return_softmax_lse, block_table)
    return (out, lse) if return_softmax_lse else out


def flash_attn_with_kvcache(
    q,
    k_cache,
    v_cache,
    cache_seqlens: Optional[Union[int, torch.Tensor]] = None,
    block_table: Optional[torch.Tensor] = None,
    num_splits=0,
    causal: bool = False,
    k: Optional[torch.Tensor] = None,
    v: Optional[torch.Tensor] = None,
    window_size=(-1, -1),
    return_softmax_lse: bool = False,
):
    """Attention against a KV cache (flash-decoding: split-KV + LSE combine for single-token steps).

    q: (batch, seqlen_q, nheads, headdim); k_cache, v_cache: (batch, seqlen_k, nheads_k, headdim), or with
    ``block_table`` (batch, max_blocks) int32: (num_blocks, page_block_size, nheads_k, headdim).
    cache_seqlens: (batch,) int32 valid lengths; None = whole cache; int = same length for every row.
    num_splits: 0 = choose automatically, 1 = no split, n = split the keys n ways.  The automatic choice is made from the
    cache's capacity and the batch size (the lengths are device data the launch never reads), which is right for caches that
    are about evenly filled; for a batch whose lengths are very uneven -- one long sequence among short ones -- pass 4 to 8:
    every sequence is then cut n ways, so no single workgroup is left streaming a whole long sequence (measured: 102 -> 45 us
    for one 8192-key sequence beside 63 of 512 keys, against +2 % on an even, full cache).
    Supersets of the reference (which supports seqlen_q == 1 without append only):
      k, v: (batch, seqlen_new, nheads_k, headdim) new tokens, written into the cache at cache_seqlens (in place)
            before attending over cache_seqlens + seqlen_new keys (cache_seqlens itself is not modified);
      seqlen_q > 1: the queries are the LAST seqlen_q positions; ``causal`` then masks bottom-right aligned;
      window_size, return_softmax_lse as in flash_attn_func.
    """
    if isinstance(cache_seqlens, int):
        cache_seqlens = torch.full((q.size(0),), cache_seqlens, dtype=torch.int32, device=q.device)
    extras = k is not None or v is not None or window_size != (-1, -1) or return_softmax_lse or q.size(1) != 1
    if not extras:
        return _C.mini_flash_attention_with_kvcache(q, k_cache, v_cache, cache_seqlens, block_table, False, num_splits)
    left, right = _window(window_size)
    out, lse = _C.kvcache_ex(q, k_cache, v_cache, k, v, cache_seqlens, block_table, causal, left, right, num_splits,
                             return_softmax_lse)
    return (out, lse) if return_softmax_lse else out
