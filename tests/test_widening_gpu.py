"""GPU parity for the SURVEY.md §8(f) "next" rows built on top of the hot path (opt-in supersets of the reference):
sliding-window attention, the LSE as an output, kv-cache attention with seqlen_q > 1 (bottom-right causal) and
kv-cache append.  Oracle: torch SDPA fp32 with an explicit boolean mask / logsumexp of the masked scaled scores."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

import hip_path as hp
from conftest import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rnd(*shape, dtype=torch.float16, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randn(*shape, generator=g).to(dtype).to(DEV)


def keep_mask(sq, sk, causal, window, bottom_right):
    """(sq, sk) bool: key t visible from row r iff r+off-left <= t <= r+off+right (and <= r+off if causal)."""
    off = sk - sq if bottom_right else 0
    r = torch.arange(sq, device=DEV)[:, None] + off
    t = torch.arange(sk, device=DEV)[None, :]
    left, right = window
    keep = torch.ones(sq, sk, dtype=torch.bool, device=DEV)
    if causal:
        keep &= t <= r
    if left >= 0:
        keep &= t >= r - left
    if right >= 0:
        keep &= t <= r + right
    return keep


def masked_ref(q, k, v, keep):
    """q (B,Sq,H,D), k/v (B,Sk,Hk,D), keep (Sq,Sk) -> out (B,Sq,H,D) fp32 with 0 rows where nothing is visible, lse (B,H,Sq)."""
    g = q.size(2) // k.size(2)
    qf, kf, vf = (x.float().transpose(1, 2) for x in (q, k, v))
    kf, vf = kf.repeat_interleave(g, dim=1), vf.repeat_interleave(g, dim=1)
    s = torch.einsum("bhqd,bhkd->bhqk", qf, kf) / (q.size(-1) ** 0.5)
    s = s.masked_fill(~keep, float("-inf"))
    lse = torch.logsumexp(s, dim=-1)
    p = torch.softmax(s, dim=-1).nan_to_num(0.0)
    return torch.einsum("bhqk,bhkd->bhqd", p, vf).transpose(1, 2), lse


@pytest.mark.parametrize("window", [(0, 0), (16, 0), (64, 0), (100, 7), (0, 50), (-1, 30), (200, -1), (1000, 1000)])
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_sliding_window_dense(mfa, window, causal, dtype):
    B, Sq, Sk, H, Hk, D = 2, 300, 300, 4, 2, 128
    q, k, v = rnd(B, Sq, H, D, dtype=dtype, seed=1), rnd(B, Sk, Hk, D, dtype=dtype, seed=2), rnd(B, Sk, Hk, D, dtype=dtype, seed=3)
    out, lse = mfa.flash_attn_func(q, k, v, causal=causal, window_size=window, return_softmax_lse=True)
    ref, ref_lse = masked_ref(q, k, v, keep_mask(Sq, Sk, causal, window, False))
    assert_close(out, ref, p_rounded=True, what=f"window={window} causal={causal}")
    fin = torch.isfinite(ref_lse)
    assert torch.equal(torch.isfinite(lse), fin)
    assert torch.allclose(lse[fin], ref_lse[fin], atol=2e-3, rtol=1e-4)


@pytest.mark.parametrize("Sq,Sk", [(64, 300), (300, 64), (129, 513), (1, 257)])
def test_sliding_window_cross_lengths_and_empty_rows(mfa, Sq, Sk):
    """Rows whose window holds no key (Sq > Sk, top-left) come out as 0 with LSE = -inf, never NaN."""
    q, k, v = rnd(1, Sq, 4, 64, seed=1), rnd(1, Sk, 4, 64, seed=2), rnd(1, Sk, 4, 64, seed=3)
    for window in ((32, 0), (10, 10), (0, 0)):
        out, lse = mfa.flash_attn_func(q, k, v, window_size=window, return_softmax_lse=True)
        ref, ref_lse = masked_ref(q, k, v, keep_mask(Sq, Sk, False, window, False))
        assert_close(out, ref, what=f"{Sq}x{Sk} window={window}")
        assert torch.equal(torch.isfinite(lse), torch.isfinite(ref_lse))


def test_lse_output_plain_and_varlen(mfa, oracle):
    """LSE = ln sum exp(scale * s) per (batch, head, row); dense layout (B,H,Sq), varlen layout (H,total_q)."""
    q, k, v = rnd(2, 200, 6, 128, seed=1), rnd(2, 333, 2, 128, seed=2), rnd(2, 333, 2, 128, seed=3)
    for causal in (False, True):
        out, lse = mfa.flash_attn_func(q, k, v, causal=causal, return_softmax_lse=True)
        ref, ref_lse = masked_ref(q, k, v, keep_mask(200, 333, causal, (-1, -1), False))
        assert_close(out, ref, what="dense out")
        assert torch.allclose(lse, ref_lse, atol=2e-3, rtol=1e-4)
        assert torch.equal(out, mfa.flash_attn_func(q, k, v, causal=causal))  # same kernel, same bits
    lens = [5, 130, 64, 257]
    cu = torch.tensor([0] + lens).cumsum(0).int().to(DEV)
    q, k, v = rnd(sum(lens), 4, 64, seed=4), rnd(sum(lens), 4, 64, seed=5), rnd(sum(lens), 4, 64, seed=6)
    out, lse = mfa.flash_attn_varlen_func(q, k, v, cu, cu, max(lens), max(lens), causal=True, return_softmax_lse=True)
    assert lse.shape == (4, sum(lens))
    for b, (a0, a1) in enumerate(zip(cu[:-1].tolist(), cu[1:].tolist())):
        ref, ref_lse = masked_ref(q[a0:a1][None], k[a0:a1][None], v[a0:a1][None], keep_mask(a1 - a0, a1 - a0, True, (-1, -1), False))
        assert_close(out[a0:a1], ref[0], what=f"varlen seq {b}")
        assert torch.allclose(lse[:, a0:a1], ref_lse[0], atol=2e-3, rtol=1e-4)
    # varlen + window
    out = mfa.flash_attn_varlen_func(q, k, v, cu, cu, max(lens), max(lens), causal=True, window_size=(40, 0))
    for a0, a1 in zip(cu[:-1].tolist(), cu[1:].tolist()):
        ref, _ = masked_ref(q[a0:a1][None], k[a0:a1][None], v[a0:a1][None], keep_mask(a1 - a0, a1 - a0, True, (40, 0), False))
        assert_close(out[a0:a1], ref[0], what="varlen window")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("Sq", [2, 5, 64, 130])
def test_kvcache_multi_query_bottom_right_causal(mfa, dtype, Sq):
    """seqlen_q > 1 against a cache: the queries are the last Sq positions; causal is bottom-right aligned (flash-attn)."""
    B, Sk, H, Hk, D = 3, 700, 8, 2, 128
    q, kc, vc = rnd(B, Sq, H, D, dtype=dtype, seed=1), rnd(B, Sk, Hk, D, dtype=dtype, seed=2), rnd(B, Sk, Hk, D, dtype=dtype, seed=3)
    lens = torch.tensor([700, max(Sq, 133), 401], dtype=torch.int32, device=DEV)
    for causal in (False, True):
        out, lse = mfa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, causal=causal, return_softmax_lse=True)
        for b in range(B):
            n = int(lens[b])
            ref, ref_lse = masked_ref(q[b:b + 1], kc[b:b + 1, :n], vc[b:b + 1, :n], keep_mask(Sq, n, causal, (-1, -1), True))
            assert_close(out[b:b + 1], ref, p_rounded=True, what=f"Sq={Sq} b={b} causal={causal}")
            assert torch.allclose(lse[b], ref_lse[0], atol=2e-3, rtol=1e-4)
    # paged cache, page 16 (below the tile) and 256
    for page in (16, 256):
        kp, vp, table = hp.make_paged(kc, vc, page, seed=page)
        out = mfa.flash_attn_with_kvcache(q, kp, vp, cache_seqlens=lens, block_table=table, causal=True)
        for b in range(B):
            n = int(lens[b])
            ref, _ = masked_ref(q[b:b + 1], kc[b:b + 1, :n], vc[b:b + 1, :n], keep_mask(Sq, n, True, (-1, -1), True))
            assert_close(out[b:b + 1], ref, p_rounded=True, what=f"paged {page} Sq={Sq} b={b}")


@pytest.mark.parametrize("paged", [False, True])
@pytest.mark.parametrize("Sn", [1, 3, 64])
def test_kvcache_append_then_attend(mfa, capi, paged, Sn):
    """k=, v= are written into the cache at cache_seqlens (in place) and attended together with the old keys."""
    B, Sk, H, Hk, D, page = 3, 512, 6, 2, 128, 64
    dtype = torch.bfloat16
    kc, vc = rnd(B, Sk, Hk, D, dtype=dtype, seed=2), rnd(B, Sk, Hk, D, dtype=dtype, seed=3)
    kn, vn = rnd(B, Sn, Hk, D, dtype=dtype, seed=4), rnd(B, Sn, Hk, D, dtype=dtype, seed=5)
    q = rnd(B, Sn, H, D, dtype=dtype, seed=1)
    lens = torch.tensor([0, 200, Sk - Sn], dtype=torch.int32, device=DEV)
    # expected cache after the append
    ke, ve = kc.clone(), vc.clone()
    for b in range(B):
        ke[b, int(lens[b]):int(lens[b]) + Sn], ve[b, int(lens[b]):int(lens[b]) + Sn] = kn[b], vn[b]
    if paged:
        kp, vp, table = hp.make_paged(kc, vc, page, seed=1)
        out = mfa.flash_attn_with_kvcache(q, kp, vp, cache_seqlens=lens, block_table=table, causal=True, k=kn, v=vn)
        got_k = kp[table.long()].reshape(B, -1, Hk, D)[:, :Sk]
        got_v = vp[table.long()].reshape(B, -1, Hk, D)[:, :Sk]
    else:
        kd, vd = kc.clone(), vc.clone()
        out = mfa.flash_attn_with_kvcache(q, kd, vd, cache_seqlens=lens, causal=True, k=kn, v=vn)
        got_k, got_v = kd, vd
    assert torch.equal(got_k, ke) and torch.equal(got_v, ve)      # byte-exact copy, nothing else touched
    for b in range(B):
        n = int(lens[b]) + Sn
        ref, _ = masked_ref(q[b:b + 1], ke[b:b + 1, :n], ve[b:b + 1, :n], keep_mask(Sn, n, True, (-1, -1), True))
        assert_close(out[b:b + 1], ref, p_rounded=(Sn > 1), what=f"append Sn={Sn} paged={paged} b={b}")


def test_kvcache_append_through_c_abi_and_capacity(capi):
    """mfa_kvcache_append called as a foreign host would; rows past the cache capacity are dropped, not written."""
    B, Sk, Hk, D, Sn = 2, 128, 2, 64, 8
    kc, vc = rnd(B, Sk, Hk, D, seed=1), rnd(B, Sk, Hk, D, seed=2)
    kn, vn = rnd(B, Sn, Hk, D, seed=3), rnd(B, Sn, Hk, D, seed=4)
    k0, v0 = kc.clone(), vc.clone()
    lens = torch.tensor([10, Sk - 3], dtype=torch.int32, device=DEV)   # second row overflows by 5
    p = capi.KvAppendParams()
    p.k_new, p.v_new, p.k_cache, p.v_cache = kn.data_ptr(), vn.data_ptr(), kc.data_ptr(), vc.data_ptr()
    for name, t in (("kn", kn), ("vn", vn), ("kc", kc), ("vc", vc)):
        setattr(p, f"{name}_batch_stride", t.stride(0)); setattr(p, f"{name}_row_stride", t.stride(1)); setattr(p, f"{name}_head_stride", t.stride(2))
    p.seqlens_k = lens.data_ptr()
    p.batch, p.seqlen_new, p.kv_heads, p.head_dim, p.seqlen_k = B, Sn, Hk, D, Sk
    rc = capi.load().mfa_kvcache_append(ctypes.byref(p), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, capi.last_error()
    torch.cuda.synchronize()
    k0[0, 10:18], v0[0, 10:18] = kn[0], vn[0]
    k0[1, Sk - 3:], v0[1, Sk - 3:] = kn[1, :3], vn[1, :3]
    assert torch.equal(kc, k0) and torch.equal(vc, v0)


def test_decode_with_window_and_defaults_unchanged(mfa):
    """A window on a single-token step goes through the MFMA path; without extras the decode kernel is used and the
    results of the two routes agree."""
    q, kc, vc = rnd(2, 1, 8, 128, seed=1), rnd(2, 600, 2, 128, seed=2), rnd(2, 600, 2, 128, seed=3)
    lens = torch.tensor([600, 77], dtype=torch.int32, device=DEV)
    base = mfa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens)
    big = mfa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, window_size=(100000, 0))
    assert (base.float() - big.float()).abs().max() < 2e-3
    win = mfa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, window_size=(31, 0))
    for b in range(2):
        n = int(lens[b])
        ref, _ = masked_ref(q[b:b + 1], kc[b:b + 1, :n], vc[b:b + 1, :n], keep_mask(1, n, False, (31, 0), True))
        assert_close(win[b:b + 1], ref, what="decode window")
