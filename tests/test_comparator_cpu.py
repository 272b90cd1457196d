"""The flash_attn comparator (testsupport/flash_attn) pinned against the oracle on the CPU, so that
the GPU tests and benchmarks that use it as "the official library" stand on checked ground."""
import os
import sys

import pytest
import torch

from conftest import ROOT
from oracle import oracle as orc

sys.path.insert(0, os.path.join(ROOT, "testsupport"))
import flash_attn as fa  # noqa: E402


def rnd(*shape, seed=0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g).to(dtype)


@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("hq,hk", [(4, 4), (8, 2), (6, 1)])
def test_dense_equals_sdpa_oracle(causal, hq, hk):
    q, k, v = rnd(2, 70, hq, 64, seed=1), rnd(2, 70, hk, 64, seed=2), rnd(2, 70, hk, 64, seed=3)
    out, lse, _ = fa.flash_attn_func(q, k, v, causal=causal, return_attn_probs=True)
    torch.testing.assert_close(out, orc.sdpa_dense(q, k, v, causal), atol=2e-6, rtol=1e-5)
    # LSE = logsumexp of the scaled, masked scores
    s = torch.einsum("bqhd,bkhd->bhqk", q, k.repeat_interleave(hq // hk, dim=2)) / 8.0
    if causal:
        s = s.masked_fill(torch.triu(torch.ones(70, 70, dtype=torch.bool), 1), float("-inf"))
    torch.testing.assert_close(lse, torch.logsumexp(s, -1), atol=1e-5, rtol=1e-5)


def test_causal_is_bottom_right_when_lengths_differ():
    q, k, v = rnd(1, 3, 2, 32, seed=4), rnd(1, 10, 2, 32, seed=5), rnd(1, 10, 2, 32, seed=6)
    out = fa.flash_attn_func(q, k, v, causal=True)
    # row i sees keys <= i + 7: the last row sees everything
    torch.testing.assert_close(out[:, 2:], orc.sdpa_dense(q[:, 2:], k, v, False), atol=2e-6, rtol=1e-5)
    torch.testing.assert_close(out[:, :1], orc.sdpa_dense(q[:, :1], k[:, :8], v[:, :8], False), atol=2e-6, rtol=1e-5)
    # more queries than keys: the leading rows see nothing -> 0 / -inf
    o2, l2, _ = fa.flash_attn_func(k, q, q, causal=True, return_attn_probs=True)
    assert (o2[:, :7] == 0).all() and torch.isinf(l2[:, :, :7]).all() and torch.isfinite(l2[:, :, 7:]).all()


def test_sliding_window():
    q, k, v = rnd(1, 40, 2, 32, seed=7), rnd(1, 40, 2, 32, seed=8), rnd(1, 40, 2, 32, seed=9)
    out = fa.flash_attn_func(q, k, v, window_size=(5, 2))
    for i in (0, 7, 39):
        lo, hi = max(0, i - 5), min(39, i + 2)
        ref = orc.sdpa_dense(q[:, i:i + 1], k[:, lo:hi + 1], v[:, lo:hi + 1], False)
        torch.testing.assert_close(out[:, i:i + 1], ref, atol=2e-6, rtol=1e-5)


@pytest.mark.parametrize("causal", [False, True])
def test_varlen_and_paged_equal_sdpa_oracle(causal):
    lens = [5, 64, 1, 33]
    cu = torch.tensor([0] + lens).cumsum(0).int()
    q, k, v = rnd(sum(lens), 4, 64, seed=10), rnd(sum(lens), 2, 64, seed=11), rnd(sum(lens), 2, 64, seed=12)
    out = fa.flash_attn_varlen_func(q, k, v, cu, cu, max(lens), max(lens), causal=causal)
    torch.testing.assert_close(out, orc.sdpa_varlen(q, k, v, cu, cu, causal), atol=2e-6, rtol=1e-5)
    # the same keys scattered into pages of 16 behind a permuted block table
    page, nblk = 16, [(n + 15) // 16 for n in lens]
    table = torch.full((4, max(nblk)), 0, dtype=torch.int32)
    perm = torch.randperm(sum(nblk), generator=torch.Generator().manual_seed(13)).tolist()
    kp, vp = torch.zeros(sum(nblk), page, 2, 64), torch.zeros(sum(nblk), page, 2, 64)
    it = iter(perm)
    for b, n in enumerate(lens):
        for j in range(nblk[b]):
            blk = next(it)
            table[b, j] = blk
            rows = slice(int(cu[b]) + j * page, int(cu[b]) + min(n, (j + 1) * page))
            kp[blk, : rows.stop - rows.start], vp[blk, : rows.stop - rows.start] = k[rows], v[rows]
    outp = fa.flash_attn_varlen_func(q, kp, vp, cu, cu, max(lens), max(lens), causal=causal, block_table=table)
    torch.testing.assert_close(outp, out, atol=1e-6, rtol=1e-6)


def test_kvcache_decode_append_and_lse():
    B, Sk, Hq, Hk, D = 3, 50, 4, 2, 32
    q, kc, vc = rnd(B, 1, Hq, D, seed=14), rnd(B, Sk, Hk, D, seed=15), rnd(B, Sk, Hk, D, seed=16)
    lens = torch.tensor([50, 17, 1], dtype=torch.int32)
    out, lse = fa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, return_softmax_lse=True)
    torch.testing.assert_close(out, orc.sdpa_decode(q, kc, vc, lens), atol=2e-6, rtol=1e-5)
    assert lse.shape == (B, Hq, 1)
    # causal with one query = non-causal (bottom-right: it is the last position)
    torch.testing.assert_close(fa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, causal=True), out, atol=0, rtol=0)
    # append two new tokens, then attend with Sq = 2, causal: row 0 must not see the second new key
    kn, vn, q2 = rnd(B, 2, Hk, D, seed=17), rnd(B, 2, Hk, D, seed=18), rnd(B, 2, Hq, D, seed=19)
    lens2 = torch.tensor([10, 17, 0], dtype=torch.int32)
    kc2, vc2 = kc.clone(), vc.clone()
    o2 = fa.flash_attn_with_kvcache(q2, kc2, vc2, k=kn, v=vn, cache_seqlens=lens2, causal=True)
    for b in range(B):
        n = int(lens2[b])
        assert torch.equal(kc2[b, n:n + 2], kn[b]) and torch.equal(vc2[b, n:n + 2], vn[b])
        assert torch.equal(kc2[b, n + 2:], kc[b, n + 2:])
        ref0 = orc.sdpa_decode(q2[b:b + 1, :1], kc2[b:b + 1], vc2[b:b + 1], torch.tensor([n + 1], dtype=torch.int32))
        ref1 = orc.sdpa_decode(q2[b:b + 1, 1:], kc2[b:b + 1], vc2[b:b + 1], torch.tensor([n + 2], dtype=torch.int32))
        torch.testing.assert_close(o2[b:b + 1, :1], ref0, atol=2e-6, rtol=1e-5)
        torch.testing.assert_close(o2[b:b + 1, 1:], ref1, atol=2e-6, rtol=1e-5)


def test_unmodelled_features_raise():
    q = rnd(1, 4, 1, 32)
    with pytest.raises(NotImplementedError):
        fa.flash_attn_func(q, q, q, dropout_p=0.1)
    with pytest.raises(NotImplementedError):
        fa.flash_attn_func(q, q, q, alibi_slopes=torch.ones(1))
    with pytest.raises(NotImplementedError):
        fa.flash_attn_with_kvcache(q, q, q, rotary_cos=torch.ones(1))
    fa.flash_attn_func(q, q, q, dropout_p=0.0, softcap=0.0, deterministic=True)
