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


def flash_attn_func(
    q: torch.Tensor,
    k: torch.Tensor,
    v: torch.Tensor,
    causal: bool = False,
) -> torch.Tensor:
    """Dense attention forward, O = softmax(Q K^T / sqrt(D) + mask) V.

    q: (batch, seqlen_q, nheads, headdim); k, v: (batch, seqlen_k, nheads_k, headdim), fp16 or bf16 on
    the GPU, nheads % nheads_k == 0 (MQA/GQA: query head h uses KV head h // (nheads // nheads_k)).
    causal: top-left aligned mask (key index > query index is masked), as the reference kernel and
    torch SDPA ``is_causal=True``.
    Returns (batch, seqlen_q, nheads, headdim) in q's dtype.
    """
    return _C.mini_flash_attention_forward(q, k, v, None, causal, -1, -1)


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
) -> torch.Tensor:
    """Packed variable-length attention forward (continuous batching).

    q: (total_q, nheads, headdim); k, v: (total_k, nheads_k, headdim), or with ``block_table``
    (batch, max_blocks) int32: paged (num_blocks, page_block_size, nheads_k, headdim).
    cu_seqlens_q / cu_seqlens_k: (batch + 1,) int32 cumulative lengths.
    Returns (total_q, nheads, headdim).
    """
    return _C.mini_flash_attention_varlen_forward(
        q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, causal, -1, -1, block_table
    )


def flash_attn_with_kvcache(
    q,
    k_cache,
    v_cache,
    cache_seqlens: Optional[Union[int, torch.Tensor]] = None,
    block_table: Optional[torch.Tensor] = None,
    num_splits=0,
    causal: bool = False,
) -> torch.Tensor:
    """Single-token decode against a KV cache (flash-decoding: split-KV + LSE combine).

    q: (batch, 1, nheads, headdim); k_cache, v_cache: (batch, seqlen_k, nheads_k, headdim), or with
    ``block_table`` (batch, max_blocks) int32: (num_blocks, page_block_size, nheads_k, headdim).
    cache_seqlens: (batch,) int32 valid lengths; None = whole cache; int = same length for every row.
    num_splits: 0 = choose automatically, 1 = no split, n = split the keys n ways.
    """
    assert q.size(1) == 1, "flash_attn_with_kvcache currently only supports seqlen_q=1 for decoding"
    if isinstance(cache_seqlens, int):
        cache_seqlens = torch.full((q.size(0),), cache_seqlens, dtype=torch.int32, device=q.device)
    return _C.mini_flash_attention_with_kvcache(q, k_cache, v_cache, cache_seqlens, block_table, False, num_splits)
