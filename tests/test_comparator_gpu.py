"""The scenarios the reference checks against the official flash_attn wheel (its tests/test_varlen.py,
test_flash_decoding.py, test_minimal.py, test_both_seqlens.py, test_output_compare.py; benchmark/decode.py), run
here against the torch-math comparator in testsupport/flash_attn (pinned to the oracle by tests/test_comparator_cpu.py).
Shapes and acceptance thresholds are the reference's (max |diff| < 0.02, mean |diff| < 0.002,
tests/test_flash_decoding.py:447-448); the tighter north-star tolerance of conftest.assert_close is applied on top.
Inputs follow the reference's recipes: unit-normalised rows for varlen (tests/test_varlen.py:35-38), plain randn
for decode."""
import os
import random
import sys

import pytest
import torch
import torch.nn.functional as F

from conftest import HALF_ULP, P_ROUND_ATOL, ROOT

sys.path.insert(0, os.path.join(ROOT, "testsupport"))
import flash_attn as fa  # noqa: E402
from flash_attn.flash_attn_interface import flash_attn_with_kvcache as fa_kvcache  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda"


def reference_bar(ours, theirs, what):
    d = (ours.float() - theirs.float()).abs()
    assert d.max().item() < 0.02 and d.mean().item() < 0.002, f"{what}: max {d.max().item():.5f} mean {d.mean().item():.6f}"
    # tighter than the reference's bar: both sides are rounded to the element type, nothing else may differ
    ulp = HALF_ULP[ours.dtype]
    bound = 2e-3 + P_ROUND_ATOL[ours.dtype] + 2 * ulp * theirs.float().abs()
    assert (d <= bound).all(), f"{what}: {(d - bound).max().item():.5f} over the rounding bound"


def packed(lq, lk, hq, hk, d, dtype=torch.float16, seed=0):
    torch.manual_seed(seed)
    q, k, v = (F.normalize(torch.randn(n, h, d, device=DEV, dtype=dtype), dim=-1) for n, h in ((sum(lq), hq), (sum(lk), hk), (sum(lk), hk)))
    cu = lambda l: torch.tensor([0] + l, device=DEV, dtype=torch.int32).cumsum(0, dtype=torch.int32)
    return q, k, v, cu(lq), cu(lk), max(lq), max(lk)


VARLEN_CASES = [  # (seqlens, q heads, kv heads, head dim, causal) -- tests/test_varlen.py:55-59,102-106,149-153,193-197,238-242,330-334
    ([512] * 4, 8, 8, 64, False), ([128, 256, 512, 1024, 64], 12, 12, 128, False), ([256, 512, 128, 1024], 16, 16, 64, True),
    ([200, 400, 600], 24, 8, 128, False), ([16, 32, 48, 8], 8, 8, 64, False), ([128, 256, 512], 8, 8, 128, False),
]


@pytest.mark.parametrize("lens,hq,hk,d,causal", VARLEN_CASES)
def test_varlen_scenarios(mfa, lens, hq, hk, d, causal):
    q, k, v, cuq, cuk, mq, mk = packed(lens, lens, hq, hk, d)
    ours = mfa.flash_attn_varlen_func(q, k, v, cuq, cuk, mq, mk, causal=causal)
    theirs = fa.flash_attn_varlen_func(q, k, v, cuq, cuk, mq, mk, causal=causal)
    reference_bar(ours, theirs, f"varlen {lens} {hq}/{hk} D{d} causal={causal}")


def test_varlen_random_lengths(mfa):
    random.seed(42)  # tests/test_varlen.py:284-289: 16 sequences of 64..512 tokens
    lens = [random.randint(64, 512) for _ in range(16)]
    q, k, v, cuq, cuk, mq, mk = packed(lens, lens, 8, 8, 64, seed=1)
    reference_bar(mfa.flash_attn_varlen_func(q, k, v, cuq, cuk, mq, mk), fa.flash_attn_varlen_func(q, k, v, cuq, cuk, mq, mk), "varlen random")


def test_varlen_uniform_equals_fixed_length(mfa):
    """tests/test_varlen.py:55-98: four equal sequences must match the fixed-length entry point."""
    q, k, v, cuq, cuk, mq, mk = packed([512] * 4, [512] * 4, 8, 8, 64, seed=2)
    a = mfa.flash_attn_varlen_func(q, k, v, cuq, cuk, mq, mk)
    b = mfa.flash_attn_func(q.view(4, 512, 8, 64), k.view(4, 512, 8, 64), v.view(4, 512, 8, 64))
    assert torch.equal(a.view(4, 512, 8, 64), b)


def paged_cache(batch, seqlen, heads, d, page, dtype, seed):
    """Identity block table, zero-filled tail of the last page (tests/test_both_seqlens.py:19-42)."""
    torch.manual_seed(seed)
    nb = (seqlen + page - 1) // page
    kc = torch.zeros(batch * nb, page, heads, d, device=DEV, dtype=dtype)
    vc = torch.zeros_like(kc)
    table = torch.arange(batch * nb, device=DEV, dtype=torch.int32).view(batch, nb)
    for b in range(batch):
        for i in range(nb):
            n = min(page, seqlen - i * page)
            kc[b * nb + i, :n] = torch.randn(n, heads, d, device=DEV, dtype=dtype)
            vc[b * nb + i, :n] = torch.randn(n, heads, d, device=DEV, dtype=dtype)
    return kc, vc, table


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("batch,seqlen,heads,d,page", [(4, 512, 8, 128, 256), (2, 512, 8, 64, 256), (1, 257, 8, 128, 256),
                                                       (4, 256, 8, 128, 256), (4, 257, 8, 128, 256), (4, 300, 4, 128, 256)])
def test_paged_decode_scenarios(mfa, dtype, batch, seqlen, heads, d, page):
    kc, vc, table = paged_cache(batch, seqlen, heads, d, page, dtype, seed=12345)
    torch.manual_seed(999)
    q = torch.randn(batch, 1, heads, d, device=DEV, dtype=dtype)
    lens = torch.full((batch,), seqlen, dtype=torch.int32, device=DEV)
    for causal in (False, True):  # one query at the last position: causal changes nothing (test_flash_decoding.py:269-326)
        ours = mfa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, block_table=table, causal=causal)
        theirs = fa_kvcache(q, kc, vc, cache_seqlens=lens, block_table=table, causal=causal)
        reference_bar(ours, theirs, f"paged decode B{batch} S{seqlen} H{heads} D{d} causal={causal}")
    # every batch element alone gives the same rows (tests/test_output_compare.py)
    one = mfa.flash_attn_with_kvcache(q[1:2] if batch > 1 else q, kc, vc, cache_seqlens=lens[:1], block_table=table[1:2] if batch > 1 else table)
    assert torch.equal(one[0], mfa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, block_table=table)[1 if batch > 1 else 0])


@pytest.mark.parametrize("seqlen_kv", [256, 512, 1024, 2048])
def test_decode_lengths_and_splits(mfa, seqlen_kv):
    """tests/test_flash_decoding.py:329-389: dense-in-pages cache, num_splits = 2 above 1024 keys else automatic."""
    kc, vc, table = paged_cache(2, seqlen_kv, 8, 128, 256, torch.float16, seed=3)
    q = torch.randn(2, 1, 8, 128, device=DEV, dtype=torch.float16)
    lens = torch.full((2,), seqlen_kv, dtype=torch.int32, device=DEV)
    ours = mfa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, block_table=table, num_splits=2 if seqlen_kv > 1024 else 0)
    reference_bar(ours, fa_kvcache(q, kc, vc, cache_seqlens=lens, block_table=table), f"decode Skv={seqlen_kv}")


@pytest.mark.parametrize("head_dim", [64, 128, 256])
def test_decode_head_dims(mfa, head_dim):
    kc, vc, table = paged_cache(2, 512, 8, head_dim, 256, torch.float16, seed=4)
    q = torch.randn(2, 1, 8, head_dim, device=DEV, dtype=torch.float16)
    lens = torch.full((2,), 512, dtype=torch.int32, device=DEV)
    reference_bar(mfa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, block_table=table),
                  fa_kvcache(q, kc, vc, cache_seqlens=lens, block_table=table), f"decode D={head_dim}")


def test_benchmark_decode_shape_with_lse(mfa):
    """benchmark/decode.py: fp16 B96 H48 Skv4096 dense cache; it unpacks (out, lse) from the official call."""
    torch.manual_seed(0)
    B, S, H, D = 96, 4096, 48, 128
    q = torch.randn(B, 1, H, D, device=DEV, dtype=torch.float16)
    kc, vc = torch.randn(B, S, H, D, device=DEV, dtype=torch.float16), torch.randn(B, S, H, D, device=DEV, dtype=torch.float16)
    lens = torch.full((B,), S, dtype=torch.int32, device=DEV)
    ours, lse = mfa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, return_softmax_lse=True)
    theirs, lse_ref = fa_kvcache(q, kc, vc, cache_seqlens=lens, return_softmax_lse=True)
    reference_bar(ours, theirs, "benchmark/decode.py shape")
    torch.testing.assert_close(lse.view_as(lse_ref), lse_ref, atol=2e-3, rtol=1e-4)


def test_generation_loop_with_append(mfa):
    """tests/test_flash_decoding.py:520-630 appends each new token to the paged cache in Python and decodes; here the
    library's own append (k=, v=) does the write and the comparator's in-place append is the counterpart."""
    torch.manual_seed(5)
    B, H, D, page, steps, start = 4, 8, 128, 256, 6, 254  # crosses the 256-key page boundary
    nb = 2
    kc = torch.zeros(B * nb, page, H, D, device=DEV, dtype=torch.float16)
    vc = torch.zeros_like(kc)
    table = torch.arange(B * nb, device=DEV, dtype=torch.int32).view(B, nb)
    kc.view(B, nb * page, H, D)[:, :start] = torch.randn(B, start, H, D, device=DEV, dtype=torch.float16)
    vc.view(B, nb * page, H, D)[:, :start] = torch.randn(B, start, H, D, device=DEV, dtype=torch.float16)
    kc2, vc2 = kc.clone(), vc.clone()
    for t in range(steps):
        q = torch.randn(B, 1, H, D, device=DEV, dtype=torch.float16)
        kn, vn = torch.randn(B, 1, H, D, device=DEV, dtype=torch.float16), torch.randn(B, 1, H, D, device=DEV, dtype=torch.float16)
        lens = torch.full((B,), start + t, dtype=torch.int32, device=DEV)
        ours = mfa.flash_attn_with_kvcache(q, kc, vc, k=kn, v=vn, cache_seqlens=lens, block_table=table)
        theirs = fa_kvcache(q, kc2, vc2, k=kn, v=vn, cache_seqlens=lens, block_table=table)
        reference_bar(ours, theirs, f"generation step {t}")
        assert torch.equal(kc, kc2) and torch.equal(vc, vc2)


@pytest.mark.parametrize("window", [(-1, -1), (128, 0), (64, 64)])
@pytest.mark.parametrize("causal", [False, True])
def test_dense_with_windows_and_lse(mfa, causal, window):
    torch.manual_seed(6)
    q, k, v = (torch.randn(2, 384, 8, 128, device=DEV, dtype=torch.bfloat16) for _ in range(3))
    ours, lse = mfa.flash_attn_func(q, k, v, causal=causal, window_size=window, return_softmax_lse=True)
    theirs, lse_ref, _ = fa.flash_attn_func(q, k, v, causal=causal, window_size=window, return_attn_probs=True)
    reference_bar(ours, theirs, f"dense causal={causal} window={window}")
    torch.testing.assert_close(lse, lse_ref, atol=2e-3, rtol=1e-4)


def test_speculative_queries_bottom_right(mfa):
    """Sq = 5 draft tokens on a cache: flash-attn aligns causal to the last key; so does the kv-cache entry here."""
    torch.manual_seed(7)
    B, Sq, Sk, Hq, Hk, D = 3, 5, 700, 16, 4, 128
    q = torch.randn(B, Sq, Hq, D, device=DEV, dtype=torch.float16)
    kc, vc = torch.randn(B, Sk, Hk, D, device=DEV, dtype=torch.float16), torch.randn(B, Sk, Hk, D, device=DEV, dtype=torch.float16)
    lens = torch.tensor([700, 64, 5], dtype=torch.int32, device=DEV)
    ours = mfa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, causal=True)
    reference_bar(ours, fa_kvcache(q, kc, vc, cache_seqlens=lens, causal=True), "Sq=5 bottom-right")
