"""GPU parity: flash-decoding (seqlen_q == 1, split-KV + LSE combine, dense and paged KV cache) of the HIP path
vs the oracle, the golden fixtures and SDPA; mirrors reference tests/test_flash_decoding.py (paged GQA decode,
num_splits, head dims 64/128/256, the 257 boundary, determinism, a generation loop) with values checked against
SDPA instead of the absent flash_attn package."""
import pytest
import torch

import hip_path as hp
from conftest import strict_report, assert_close, from_bits, load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rnd(*shape, dtype=torch.bfloat16, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randn(*shape, generator=g).to(dtype).to(DEV)


def sdpa_decode_gpu(q, kc, vc, lens):
    outs = []
    for b in range(q.size(0)):
        n = int(lens[b]) if lens is not None else kc.size(1)
        if n == 0:
            outs.append(torch.zeros(1, q.size(2), q.size(3), device=q.device))
        else:
            outs.append(hp.sdpa_gpu(q[b:b + 1], kc[b:b + 1, :n], vc[b:b + 1, :n])[0])
    return torch.stack(outs)


@pytest.mark.parametrize("route", hp.ROUTES)
def test_golden_g3_g5(route, mfa, capi):
    g = load_golden("g3_bf16_decode_gqa")
    q, k, v = (from_bits(g[n], torch.bfloat16).to(DEV) for n in "qkv")
    for i in range(int(g["n"])):
        lens = torch.from_numpy(g[f"lens{i}"]).to(DEV)
        for splits in (0, 1, 2, 7):
            out = hp.decode(route, mfa, capi, q, k, v, lens, num_splits=splits)
            assert_close(out, torch.from_numpy(g[f"expect{i}"]), what=f"g3[{i}] splits={splits}")
            if splits == 0:
                strict_report(out, torch.from_numpy(g[f"expect{i}"]), f"config3-shaped golden g3[{i}] bf16 GQA decode ({route})",
                              must_pass=False, why="bf16 OUTPUT rounding: half an ulp of the stored result is 2^-9*|ref| = 1.95e-3*|ref|, above the bar's 1e-3*|ref| wherever |ref| > ~1 (short caches: |ref| up to 3); the fp32 partials meet the bar (test_partials_and_lse_vs_c_restatement)")
    g = load_golden("g5_bf16_paged_decode")
    for i in range(int(g["n"])):
        q, k, v = (from_bits(g[f"{n}{i}"], torch.bfloat16).to(DEV) for n in "qkv")
        table, lens = torch.from_numpy(g[f"table{i}"]).to(DEV), torch.from_numpy(g[f"lens{i}"]).to(DEV)
        for splits in (0, 1, 3):
            out = hp.decode(route, mfa, capi, q, k, v, lens, block_table=table, num_splits=splits)
            assert_close(out, torch.from_numpy(g[f"expect{i}"]), what=f"g5[{i}] splits={splits}")
            if splits == 0:
                strict_report(out, torch.from_numpy(g[f"expect{i}"]), f"config5-shaped golden g5[{i}] bf16 paged decode ({route})",
                              must_pass=False, why="bf16 OUTPUT rounding: half an ulp of the stored result is 2^-9*|ref| = 1.95e-3*|ref|, above the bar's 1e-3*|ref| wherever |ref| > ~1 (short caches: |ref| up to 3); the fp32 partials meet the bar (test_partials_and_lse_vs_c_restatement)")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("splits", [1, 2, 5])
def test_partials_and_lse_vs_c_restatement(oracle, mfa, capi, dtype, splits):
    """Per-split fp32 partial O / LSE (the (S,B,H,D) and (S,B,H) workspaces of the reference, api.cpp:332-337) and
    the final LSE, against the CPU restatement run with the same split ranges (decode.cuh:26-30)."""
    q, kc, vc = rnd(2, 1, 6, 128, dtype=dtype, seed=1), rnd(2, 500, 2, 128, dtype=dtype, seed=2), rnd(2, 500, 2, 128, dtype=dtype, seed=3)
    lens = torch.tensor([500, 130], dtype=torch.int32, device=DEV)
    o, lse, o_acc, lse_acc, S = hp.decode("capi", mfa, capi, q, kc, vc, lens, num_splits=splits, return_partials=True)
    ro, rlse, ro_acc, rlse_acc = oracle.restated_decode(q.cpu(), kc.cpu(), vc.cpu(), lens.cpu(), num_splits=splits, return_partials=True)
    assert S == splits
    assert_close(o, oracle.sdpa_decode(q.cpu(), kc.cpu(), vc.cpu(), lens.cpu()), what="decode out")
    assert torch.allclose(lse.cpu(), rlse, atol=2e-4, rtol=1e-5)
    if S > 1:
        fin = torch.isfinite(rlse_acc)
        assert torch.equal(torch.isfinite(lse_acc.cpu()), fin)          # empty splits keep -inf (decode.cuh:533-536)
        assert torch.allclose(lse_acc.cpu()[fin], rlse_acc[fin], atol=2e-4, rtol=1e-5)
        assert torch.allclose(o_acc.cpu()[fin], ro_acc[fin], atol=2e-5, rtol=1e-4)


@pytest.mark.parametrize("D", [32, 64, 96, 128, 256])
@pytest.mark.parametrize("Hq,Hk", [(4, 4), (8, 2), (6, 2), (5, 1), (8, 1), (16, 1), (24, 8), (12, 2)])
def test_head_dims_and_group_sizes(mfa, capi, D, Hq, Hk):
    """reference tests/test_flash_decoding.py:392 head dims (+32, 96) x GQA ratios incl. groups > 8 (chunked)."""
    B, Sk = 3, 700
    q, kc, vc = rnd(B, 1, Hq, D, seed=1), rnd(B, Sk, Hk, D, seed=2), rnd(B, Sk, Hk, D, seed=3)
    lens = torch.tensor([700, 1, 389], dtype=torch.int32, device=DEV)
    ref = sdpa_decode_gpu(q, kc, vc, lens)
    for splits in (0, 1, 3):
        assert_close(hp.decode("api", mfa, capi, q, kc, vc, lens, num_splits=splits), ref, what=f"D{D} {Hq}:{Hk} splits={splits}")


@pytest.mark.parametrize("Sk", [1, 2, 63, 64, 65, 255, 256, 257, 1000, 2048, 4097])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_cache_lengths_and_split_equivalence(mfa, capi, Sk, dtype):
    """Boundaries around the 64-key tile / 256 page (reference tests: 256, 257, 512..2048 with num_splits=2);
    every split count gives the same answer as no split, up to fp32 summation order."""
    q, kc, vc = rnd(2, 1, 8, 128, dtype=dtype, seed=1), rnd(2, Sk, 2, 128, dtype=dtype, seed=2), rnd(2, Sk, 2, 128, dtype=dtype, seed=3)
    lens = torch.tensor([Sk, max(1, Sk // 2)], dtype=torch.int32, device=DEV)
    ref = sdpa_decode_gpu(q, kc, vc, lens)
    base = hp.decode("api", mfa, capi, q, kc, vc, lens, num_splits=1)
    assert_close(base, ref, what=f"Sk={Sk}")
    for splits in (0, 2, 7, 128):
        for route in hp.ROUTES:
            out = hp.decode(route, mfa, capi, q, kc, vc, lens, num_splits=splits)
            assert_close(out, ref, what=f"Sk={Sk} splits={splits} {route}")
            ulp = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
            assert ((out.float() - base.float()).abs() <= ulp * base.float().abs() + 1e-5).all()


@pytest.mark.parametrize("D", [32, 64, 96, 128, 256])
@pytest.mark.parametrize("Hq,Hk,Sq", [(4, 1, 1), (8, 1, 1), (8, 1, 3)])
def test_many_splits_small_head_dims(mfa, capi, D, Hq, Hk, Sq):
    """Split counts above head_dim/2 (the combine kernel keeps the split weights two per lane; a lane whose
    column pair lies past head_dim must still hand out its weights).  Auto (0) picks 40+ splits for B=1, few KV
    heads and a long cache; 33/64/128 are forced.  Both kv-cache routes: the vector kernel (G <= 4, Sq = 1) and
    the packed-row kernel (G = 8, or Sq > 1); every split count must agree with num_splits=1 and with SDPA."""
    B, Sk = 1, 8200
    q, kc, vc = rnd(B, Sq, Hq, D, seed=11), rnd(B, Sk, Hk, D, seed=12), rnd(B, Sk, Hk, D, seed=13)
    lens = torch.tensor([Sk], dtype=torch.int32, device=DEV)
    if Sq == 1:
        ref = sdpa_decode_gpu(q, kc, vc, lens)
    else:  # bottom-right causal over the cache (flash-attn >= 2.1), as flash_attn_with_kvcache(causal=True)
        qf = q.float().transpose(1, 2)
        kf = kc.float().transpose(1, 2).repeat_interleave(Hq // Hk, dim=1)
        vf = vc.float().transpose(1, 2).repeat_interleave(Hq // Hk, dim=1)
        sc = qf @ kf.transpose(-1, -2) / D ** 0.5
        keep = torch.arange(Sk, device=DEV)[None, :] <= (torch.arange(Sq, device=DEV)[:, None] + Sk - Sq)
        ref = (torch.softmax(sc.masked_fill(~keep, float("-inf")), -1) @ vf).transpose(1, 2)
    kw = dict(cache_seqlens=lens) if Sq == 1 else dict(cache_seqlens=lens, causal=True)
    base = mfa.flash_attn_with_kvcache(q, kc, vc, num_splits=1, **kw)
    assert_close(base, ref, what=f"D{D} G{Hq // Hk} Sq{Sq} splits=1")
    for splits in (0, 33, 64, 128):
        out = mfa.flash_attn_with_kvcache(q, kc, vc, num_splits=splits, **kw)
        assert_close(out, ref, what=f"D{D} G{Hq // Hk} Sq{Sq} splits={splits}")
        # the vector kernel keeps P in fp32, so its splits agree to summation order; the packed-row (MFMA) kernel rounds
        # P to bf16 per tile (as the prefill kernel does), so different split boundaries differ by that rounding
        packed = Sq > 1 or Hq // Hk > 4
        assert ((out.float() - base.float()).abs() <= 2.0 ** -7 * base.float().abs() + (4e-3 if packed else 1e-5)).all(), f"splits={splits}"


@pytest.mark.parametrize("page", [1, 16, 48, 64, 256])
def test_paged_cache_any_page_size(mfa, capi, page):
    """Paged decode with permuted pages; page sizes below the tile, non-powers of two and 1 are all exact."""
    B, Sk, Hq, Hk, D = 3, 530, 8, 2, 128
    q, kc, vc = rnd(B, 1, Hq, D, seed=1), rnd(B, Sk, Hk, D, seed=2), rnd(B, Sk, Hk, D, seed=3)
    kp, vp, table = hp.make_paged(kc, vc, page, seed=page)
    lens = torch.tensor([530, 77, 256], dtype=torch.int32, device=DEV)
    ref = sdpa_decode_gpu(q, kc, vc, lens)
    for splits in (0, 1, 4):
        for route in hp.ROUTES:
            assert_close(hp.decode(route, mfa, capi, q, kp, vp, lens, block_table=table, num_splits=splits), ref,
                         what=f"page={page} splits={splits} {route}")


def test_cache_seqlens_forms_and_causal_keyword(mfa, capi):
    """cache_seqlens: tensor / None (= whole cache) / int (broadcast); causal= is accepted and cannot change a
    seqlen_q == 1 result (reference tests pass it: test_flash_decoding.py:70; C++ ignores it: api.cpp:349)."""
    q, kc, vc = rnd(2, 1, 4, 64, seed=1), rnd(2, 320, 4, 64, seed=2), rnd(2, 320, 4, 64, seed=3)
    full = mfa.flash_attn_with_kvcache(q, kc, vc)
    assert_close(full, sdpa_decode_gpu(q, kc, vc, None), what="None")
    assert torch.equal(mfa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=320), full)
    part = mfa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=100)
    assert_close(part, sdpa_decode_gpu(q, kc, vc, torch.tensor([100, 100])), what="int")
    assert torch.equal(mfa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=100, causal=True), part)
    # zero-length rows give 0, not NaN
    lens = torch.tensor([0, 5], dtype=torch.int32, device=DEV)
    out = mfa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, num_splits=2)
    assert (out[0] == 0).all() and torch.isfinite(out).all()


def test_determinism_and_generation_loop(mfa, capi):
    """reference tests/test_flash_decoding.py:520-684: a 10-step generation loop appending to the paged cache, and
    repeat-run equality (1e-5 upstream; bitwise here)."""
    B, Hq, Hk, D, page = 2, 8, 2, 128, 256
    nb = 3
    kp, vp = rnd(B * nb, page, Hk, D, seed=1), rnd(B * nb, page, Hk, D, seed=2)
    table = torch.arange(B * nb, dtype=torch.int32, device=DEV).view(B, nb)
    lens = torch.tensor([250, 400], dtype=torch.int32, device=DEV)
    for step in range(10):
        q = rnd(B, 1, Hq, D, seed=100 + step)
        kn, vn = rnd(B, Hk, D, seed=200 + step), rnd(B, Hk, D, seed=300 + step)
        for b in range(B):
            pos = int(lens[b])
            kp[table[b, pos // page], pos % page], vp[table[b, pos // page], pos % page] = kn[b], vn[b]
        lens += 1
        out = mfa.flash_attn_with_kvcache(q, kp, vp, cache_seqlens=lens, block_table=table)
        assert torch.equal(out, mfa.flash_attn_with_kvcache(q, kp, vp, cache_seqlens=lens, block_table=table))
        kd = kp.view(B, nb * page, Hk, D)
        vd = vp.view(B, nb * page, Hk, D)
        assert_close(out, sdpa_decode_gpu(q, kd, vd, lens), what=f"step {step}")


def test_baseline_config3_and_config5_full_size(mfa, capi):
    """BASELINE config 3 (bf16 B24 Skv8192 Hq24 Hkv8 D128, num_splits auto) and config 5 (bf16 paged B16 Skv4096
    page 256) vs SDPA-fp32 on the GPU, plus split-count invariance and page-permutation invariance."""
    B, Sk, Hq, Hk, D = 24, 8192, 24, 8, 128
    q, kc, vc = rnd(B, 1, Hq, D, seed=1), rnd(B, Sk, Hk, D, seed=2), rnd(B, Sk, Hk, D, seed=3)
    lens = torch.full((B,), Sk, dtype=torch.int32, device=DEV)
    ref = sdpa_decode_gpu(q, kc, vc, lens)
    auto = hp.decode("api", mfa, capi, q, kc, vc, lens, num_splits=0)
    assert_close(auto, ref, what="config3 auto")
    strict_report(auto, ref, "config3 full size bf16 B24 Skv8192 Hq24 Hkv8 D128 num_splits=auto")
    for splits in (1, 3, 16):
        out = hp.decode("capi", mfa, capi, q, kc, vc, lens, num_splits=splits)
        assert_close(out, ref, what=f"config3 splits={splits}")
        assert ((out.float() - auto.float()).abs() <= 2.0 ** -7 * auto.float().abs() + 1e-5).all()
    B, Sk, page = 16, 4096, 256
    kp, vp, table = hp.make_paged(kc[:B, :Sk].contiguous(), vc[:B, :Sk].contiguous(), page, seed=7, extra_blocks=0)
    lens = torch.full((B,), Sk, dtype=torch.int32, device=DEV)
    out = hp.decode("api", mfa, capi, q[:B], kp, vp, lens, block_table=table)
    assert_close(out, sdpa_decode_gpu(q[:B], kc[:B, :Sk], vc[:B, :Sk], lens), what="config5")
    strict_report(out, sdpa_decode_gpu(q[:B], kc[:B, :Sk], vc[:B, :Sk], lens), "config5 full size bf16 paged B16 Skv4096 page256")
    kp2, vp2, table2 = hp.make_paged(kc[:B, :Sk].contiguous(), vc[:B, :Sk].contiguous(), page, seed=8, extra_blocks=0)
    assert torch.equal(hp.decode("api", mfa, capi, q[:B], kp2, vp2, lens, block_table=table2), out)
