"""kv-cache attention through the packed-row MFMA kernel (mfa_run_flash_attention_with_kv_cache with seqlen_q > 1
or a GQA group > 4): the G * seqlen_q query rows of a KV head share 32-row MFMA tiles, keys are split over
workgroups and merged by the combine kernel.  Checked against the flash_attn-semantics comparator
(testsupport/flash_attn, pinned to the oracle by tests/test_comparator_cpu.py): bottom-right causal, per-batch
cache lengths (including 0 and lengths shorter than seqlen_q), paged caches, sliding windows, the LSE output,
forced split counts, row blocks (> 128 packed rows) and the three head dims with a packed instance."""
import ctypes
import os
import sys

import pytest
import torch

import hip_path as hp
from conftest import HALF_ULP, P_ROUND_ATOL, ROOT
from oracle.oracle import fill_params

sys.path.insert(0, os.path.join(ROOT, "testsupport"))
import flash_attn as fa  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rnd(*shape, dtype=torch.float16, seed=0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return torch.randn(*shape, device=DEV, dtype=torch.float32, generator=g).to(dtype)


def close(ours, theirs, what, lse=None, lse_ref=None):
    d = (ours.float() - theirs.float()).abs()
    # both sides are rounded to the element type; the MFMA path also rounds P before P.V (conftest.P_ROUND_ATOL)
    bound = 2e-3 + P_ROUND_ATOL[ours.dtype] + 2 * HALF_ULP[ours.dtype] * theirs.float().abs()
    assert torch.isfinite(ours.float()).all(), f"{what}: non-finite output"
    assert (d <= bound).all(), f"{what}: {(d - bound).max().item():.5f} over the bound (max diff {d.max().item():.5f})"
    if lse is not None:
        fin = torch.isfinite(lse_ref)
        assert torch.equal(torch.isfinite(lse), fin), f"{what}: LSE -inf pattern differs"
        torch.testing.assert_close(lse[fin], lse_ref[fin], atol=2e-3, rtol=1e-4)


LENS = [0, 1, 2, 63, 64, 65, 127, 128, 500, 1000, 1023, 1024]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("Hq,Hk", [(32, 4), (16, 2), (8, 1), (12, 2)])
@pytest.mark.parametrize("num_splits", [0, 1, 2, 7])
def test_decode_large_groups(mfa, dtype, Hq, Hk, num_splits):
    """seqlen_q = 1 with G = 8 / 6: served by the packed kernel instead of G heads in VALU registers."""
    B, Sk, D = len(LENS), 1024, 128
    q, kc, vc = rnd(B, 1, Hq, D, dtype=dtype, seed=1), rnd(B, Sk, Hk, D, dtype=dtype, seed=2), rnd(B, Sk, Hk, D, dtype=dtype, seed=3)
    lens = torch.tensor(LENS, dtype=torch.int32, device=DEV)
    ours, lse = mfa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, num_splits=num_splits, return_softmax_lse=True)
    theirs, lse_ref = fa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, return_softmax_lse=True)
    close(ours, theirs, f"G={Hq // Hk} splits={num_splits}", lse.view_as(lse_ref), lse_ref)
    assert (ours[0] == 0).all()


@pytest.mark.parametrize("Sq", [2, 5, 16, 33])
@pytest.mark.parametrize("Hq,Hk", [(8, 8), (24, 8), (32, 4)])
@pytest.mark.parametrize("causal", [False, True])
def test_few_query_tokens(mfa, Sq, Hq, Hk, causal):
    B, Sk, D = len(LENS), 1024, 128
    q, kc, vc = rnd(B, Sq, Hq, D, seed=4), rnd(B, Sk, Hk, D, seed=5), rnd(B, Sk, Hk, D, seed=6)
    lens = torch.tensor(LENS, dtype=torch.int32, device=DEV)
    ours, lse = mfa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, causal=causal, return_softmax_lse=True)
    theirs, lse_ref = fa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, causal=causal, return_softmax_lse=True)
    close(ours, theirs, f"Sq={Sq} {Hq}/{Hk} causal={causal}", lse, lse_ref)


@pytest.mark.parametrize("D", [32, 64, 96, 128, 256])
@pytest.mark.parametrize("page", [16, 48, 64, 192, 256])
def test_paged_cache_and_head_dims(mfa, D, page):
    B, Sq, Hq, Hk, Sk = 5, 4, 16, 2, 700
    q = rnd(B, Sq, Hq, D, dtype=torch.bfloat16, seed=7)
    kc, vc = rnd(B, Sk, Hk, D, dtype=torch.bfloat16, seed=8), rnd(B, Sk, Hk, D, dtype=torch.bfloat16, seed=9)
    kp, vp, table = hp.make_paged(kc, vc, page, seed=10)
    lens = torch.tensor([700, 3, 64, 333, 699], dtype=torch.int32, device=DEV)
    ours = mfa.flash_attn_with_kvcache(q, kp, vp, cache_seqlens=lens, block_table=table, causal=True)
    theirs = fa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, causal=True)
    close(ours, theirs, f"paged D={D} page={page}")


def test_row_blocks_and_windows(mfa):
    """40 query tokens x G = 8 = 320 packed rows = 3 row blocks; with and without a sliding window."""
    B, Sq, Hq, Hk, Sk, D = 3, 40, 16, 2, 2000, 128
    q, kc, vc = rnd(B, Sq, Hq, D, seed=11), rnd(B, Sk, Hk, D, seed=12), rnd(B, Sk, Hk, D, seed=13)
    lens = torch.tensor([2000, 40, 777], dtype=torch.int32, device=DEV)
    for window in ((-1, -1), (100, 0), (30, 5)):
        ours, lse = mfa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, causal=True, window_size=window, return_softmax_lse=True)
        theirs, lse_ref = fa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, causal=True, window_size=window, return_softmax_lse=True)
        close(ours, theirs, f"320 rows window={window}", lse, lse_ref)


def test_append_then_attend(mfa):
    B, Sn, Hq, Hk, Sk, D = 4, 3, 32, 4, 512, 128
    q, kn, vn = rnd(B, Sn, Hq, D, seed=14), rnd(B, Sn, Hk, D, seed=15), rnd(B, Sn, Hk, D, seed=16)
    kc, vc = rnd(B, Sk, Hk, D, seed=17), rnd(B, Sk, Hk, D, seed=18)
    kc2, vc2 = kc.clone(), vc.clone()
    lens = torch.tensor([0, 100, 509, 255], dtype=torch.int32, device=DEV)
    ours = mfa.flash_attn_with_kvcache(q, kc, vc, k=kn, v=vn, cache_seqlens=lens, causal=True)
    theirs = fa.flash_attn_with_kvcache(q, kc2, vc2, k=kn, v=vn, cache_seqlens=lens, causal=True)
    close(ours, theirs, "append + attend")
    assert torch.equal(kc, kc2) and torch.equal(vc, vc2)


def test_c_abi_plan_and_forced_splits(capi):
    """Straight through include/mfa.h: mfa_kvcache_plan sizes the workspaces; every split count gives the same
    answer to rounding; a missing workspace is refused."""
    lib = capi.load()
    B, Sq, Hq, Hk, Sk, D = 6, 4, 32, 4, 3000, 128
    q, kc, vc = rnd(B, Sq, Hq, D, dtype=torch.bfloat16, seed=19), rnd(B, Sk, Hk, D, dtype=torch.bfloat16, seed=20), rnd(B, Sk, Hk, D, dtype=torch.bfloat16, seed=21)
    lens = torch.tensor([3000, 2999, 64, 1, 1500, 0], dtype=torch.int32, device=DEV)
    ref = fa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, causal=True)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for want in (0, 1, 3, 16):
        o = torch.full_like(q, float("nan"))
        p = fill_params(q, kc, vc, o, causal=True, seqlens_k=lens)
        p.num_splits = want
        s, ob, lb = ctypes.c_int(), ctypes.c_size_t(), ctypes.c_size_t()
        assert lib.mfa_kvcache_plan(ctypes.byref(p), ctypes.byref(s), ctypes.byref(ob), ctypes.byref(lb)) == 0
        assert s.value >= 1 and (want < 1 or s.value == want)
        assert (ob.value, lb.value) == ((s.value * B * Sq * Hq * D * 4, s.value * B * Sq * Hq * 4) if s.value > 1 else (0, 0))
        p.num_splits = s.value
        if s.value > 1:
            assert lib.mfa_run_flash_attention_with_kv_cache(ctypes.byref(p), stream) == capi.MFA_ERR_WORKSPACE
            oacc = torch.empty(ob.value // 4, dtype=torch.float32, device=DEV)
            lacc = torch.empty(lb.value // 4, dtype=torch.float32, device=DEV)
            p.oaccum_ptr, p.softmax_lseaccum_ptr = oacc.data_ptr(), lacc.data_ptr()
        assert lib.mfa_run_flash_attention_with_kv_cache(ctypes.byref(p), stream) == 0, capi.last_error()
        torch.cuda.synchronize()
        close(o, ref, f"C ABI splits={s.value}")
        # without counters the merge is decode_combine_kernel's launch; with the caller's counters (and mfa_init) it runs in
        # the split kernel -- same partials, same merge function: bit-identical output, counters back at zero
        route = lib.mfa_debug_last_route()
        assert route & capi.MFA_ROUTE_PACKED and not route & capi.MFA_ROUTE_FUSED_MERGE
        assert bool(route & capi.MFA_ROUTE_COMBINE_LAUNCH) == (s.value > 1)
        if s.value > 1:
            assert lib.mfa_kvcache_counter_count(ctypes.byref(p)) == B * Hk * ((Sq * (Hq // Hk) + 127) // 128)
            assert lib.mfa_init(-1) == 1
            import hip_path as hp
            hp.split_counters(capi, p)
            o2 = torch.full_like(q, float("nan"))
            p.o_ptr = o2.data_ptr()
            for _ in range(2):
                assert lib.mfa_run_flash_attention_with_kv_cache(ctypes.byref(p), stream) == 0, capi.last_error()
                assert lib.mfa_debug_last_route() == capi.MFA_ROUTE_PACKED | capi.MFA_ROUTE_FUSED_MERGE
                torch.cuda.synchronize()
                assert torch.equal(o2, o)
            key = (torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream)
            assert int(hp._COUNTERS[key].abs().sum()) == 0


def test_strided_queries_and_caches(mfa):
    """q is a slice of a wider tensor (row / batch strides larger than the packed shape), the cache a slice of a longer
    one with more heads: the packed kernel must take every stride from the parameter block."""
    B, Sq, Hq, Hk, Sk, D = 3, 4, 16, 2, 600, 128
    q_wide = rnd(B, Sq + 2, Hq + 8, D, seed=22)
    q = q_wide[:, 1:1 + Sq, 4:4 + Hq]
    kc_wide, vc_wide = rnd(B + 1, Sk + 40, Hk + 1, D, seed=23), rnd(B + 1, Sk + 40, Hk + 1, D, seed=24)
    kc, vc = kc_wide[1:, 8:8 + Sk, :Hk], vc_wide[1:, 8:8 + Sk, 1:]
    assert not q.is_contiguous() and not kc.is_contiguous()
    lens = torch.tensor([600, 77, 300], dtype=torch.int32, device=DEV)
    ours = mfa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, causal=True)
    theirs = fa.flash_attn_with_kvcache(q.contiguous(), kc.contiguous(), vc.contiguous(), cache_seqlens=lens, causal=True)
    close(ours, theirs, "strided q / cache")


@pytest.mark.parametrize("causal", [True, False])
def test_long_query_block_on_few_heads_takes_the_packed_kernel(mfa, capi, causal):
    """A prompt chunk on one or two KV heads over a long cache: the per-head prefill kernel would have too few workgroups
    for the chip and cannot split the keys, so the route goes to the packed kernel (thousands of packed rows, many row blocks,
    key splits) -- values and LSE against the comparator, per-batch cache lengths included."""
    lib = capi.load()
    for B, Sq, Hq, Hk, Sk, lens in ((1, 1500, 8, 1, 4500, [4500]), (2, 700, 4, 2, 5000, [4800, 3000]), (1, 1000, 4, 4, 4100, [4100])):
        q, kc, vc = rnd(B, Sq, Hq, 128, dtype=torch.bfloat16, seed=1), rnd(B, Sk, Hk, 128, dtype=torch.bfloat16, seed=2), rnd(B, Sk, Hk, 128, dtype=torch.bfloat16, seed=3)
        cl = torch.tensor(lens, dtype=torch.int32, device=DEV)
        ours, lse = mfa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=cl, causal=causal, return_softmax_lse=True)
        assert lib.mfa_debug_last_route() & capi.MFA_ROUTE_PACKED, (B, Sq, Hq, Hk)
        theirs, lse_ref = fa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=cl, causal=causal, return_softmax_lse=True)
        close(ours, theirs, f"B{B} Sq{Sq} {Hq}/{Hk} causal={causal}", lse.view_as(lse_ref), lse_ref)
