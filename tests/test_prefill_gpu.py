"""GPU parity: prefill / varlen / paged prefill of the HIP path vs the oracle (C restatement + SDPA fp32), the
golden fixtures, and size-independent properties at BASELINE sizes.  Mirrors what reference tests/test_mha.py,
test_causal.py, test_gqa.py, test_arbitrary_seqlen.py, test_varlen.py and test_varlen_block_table.py cover (the
last one value-checked here; upstream it only checks finiteness)."""
import os

import pytest
import torch

import hip_path as hp
from conftest import strict_report, P_ROUND_ATOL, assert_close, from_bits, load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rnd(*shape, dtype=torch.float16, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randn(*shape, generator=g).to(dtype).to(DEV)


@pytest.mark.parametrize("route", hp.ROUTES)
def test_golden_g1_g2_g4(route, mfa, capi):
    g = load_golden("g1_fp32_b2_s128_h4_d64")
    for dt in (torch.float16, torch.bfloat16):
        q, k, v = (torch.from_numpy(g[n]).to(dt).to(DEV) for n in "qkv")
        out = hp.prefill(route, mfa, capi, q, k, v)
        assert_close(out, hp.sdpa_gpu(q, k, v), p_rounded=True, what=f"g1 {dt}")
        if dt == torch.float16:
            strict_report(out, hp.sdpa_gpu(q, k, v), f"config1 g1 fp16 ({route})")
        # fp32 fixture vs 16-bit inputs: input rounding dominates; the reference's own bar (test_mha.py:90-91)
        d = (out.float().cpu() - torch.from_numpy(g["expect"])).abs()
        assert d.max() < 0.02 and d.mean() < 0.002
    g = load_golden("g2_fp16_causal_d128")
    for i in range(int(g["n"])):
        q, k, v = (from_bits(g[f"{n}{i}"], torch.float16).to(DEV) for n in "qkv")
        out = hp.prefill(route, mfa, capi, q, k, v, True)
        assert_close(out, torch.from_numpy(g[f"expect{i}"]), p_rounded=True, what=f"g2[{i}]")
        strict_report(out, torch.from_numpy(g[f"expect{i}"]), f"config2-shaped golden g2[{i}] fp16 causal D128 ({route})")
    g = load_golden("g4_fp16_varlen_h8_d64")
    q, k, v = (from_bits(g[n], torch.float16).to(DEV) for n in "qkv")
    cu = torch.from_numpy(g["cu"]).to(DEV)
    out = hp.prefill(route, mfa, capi, q, k, v, True, cu_q=cu, cu_k=cu, max_sq=512, max_sk=512)
    assert_close(out, torch.from_numpy(g["expect"]), p_rounded=True, what="g4 (BASELINE config 4)")
    strict_report(out, torch.from_numpy(g["expect"]), f"config4 g4 fp16 varlen H8 D64 causal ({route})")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("D", [32, 64, 96, 128, 160, 256])
@pytest.mark.parametrize("causal", [False, True])
def test_vs_c_restatement_small(oracle, mfa, capi, dtype, D, causal):
    """Same seeded inputs through the HIP kernel and the CPU restatement of the reference algorithm."""
    B, S, H, Hk = 2, 150, 4, 2
    q, k, v = rnd(B, S, H, D, dtype=dtype, seed=1), rnd(B, S, Hk, D, dtype=dtype, seed=2), rnd(B, S, Hk, D, dtype=dtype, seed=3)
    out = hp.prefill("capi", mfa, capi, q, k, v, causal)
    rest = oracle.restated_prefill(q.cpu(), k.cpu(), v.cpu(), causal)
    ref = oracle.sdpa_dense(q.cpu(), k.cpu(), v.cpu(), causal)
    assert_close(out, ref, p_rounded=True, what="hip vs sdpa")
    # kernel and restatement agree to within two output ulps plus the P-rounding slack (fp32 summation order and
    # v_exp_f32 vs libm exp2f differ in the last bit, which can flip the 16-bit rounding of a P element)
    ulp = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    assert ((out.float().cpu() - rest.float()).abs() <= 2 * ulp * rest.float().abs() + 1e-3 + P_ROUND_ATOL[dtype]).all()


@pytest.mark.parametrize("B,S,H,D", [(1, 64, 1, 64), (2, 128, 4, 64), (4, 256, 8, 64), (8, 512, 2, 128), (2, 1024, 16, 128),
                                     (1, 2048, 4, 128), (2, 333, 4, 32), (1, 777, 2, 256), (2, 450, 4, 96)])
@pytest.mark.parametrize("causal", [False, True])
def test_mha_shapes_vs_sdpa(mfa, capi, B, S, H, D, causal):
    """reference tests/test_mha.py:109-171 and test_causal.py:119-170 parameter space (batch, seqlen, heads, dim)."""
    q, k, v = (rnd(B, S, H, D, seed=s) for s in (1, 2, 3))
    out = hp.prefill("api", mfa, capi, q, k, v, causal)
    ref = hp.sdpa_gpu(q, k, v, causal)
    d = (out.float() - ref).abs()
    assert d.max() < 0.01 and d.mean() < 0.001   # the reference's own bar
    assert_close(out, ref, p_rounded=True, what="mha")            # ours


@pytest.mark.parametrize("Hq,Hk", [(8, 1), (8, 2), (8, 4), (16, 2), (32, 8), (24, 8), (6, 2), (5, 1)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_gqa_ratios(mfa, capi, Hq, Hk, dtype):
    """reference tests/test_gqa.py:58-65 ratios (+ 24:8 of BASELINE config 3 and odd groups)."""
    q, k, v = rnd(2, 300, Hq, 128, dtype=dtype, seed=1), rnd(2, 300, Hk, 128, dtype=dtype, seed=2), rnd(2, 300, Hk, 128, dtype=dtype, seed=3)
    assert_close(hp.prefill("api", mfa, capi, q, k, v, True), hp.sdpa_gpu(q, k, v, True), p_rounded=True, what=f"gqa {Hq}:{Hk}")


@pytest.mark.parametrize("S", [1, 7, 31, 33, 63, 65, 100, 127, 129, 200, 511, 513, 1000, 2047])
def test_arbitrary_seqlen(mfa, capi, S):
    """reference tests/test_arbitrary_seqlen.py:13,77."""
    q, k, v = (rnd(2, S, 4, 128, seed=s) for s in (4, 5, 6))
    for causal in (False, True):
        assert_close(hp.prefill("capi", mfa, capi, q, k, v, causal), hp.sdpa_gpu(q, k, v, causal), p_rounded=True, what=f"S={S} causal={causal}")


@pytest.mark.parametrize("Sq,Sk", [(40, 90), (90, 40), (1, 300), (300, 1), (128, 129), (200, 64), (129, 1000)])
def test_cross_lengths_top_left_causal(mfa, capi, Sq, Sk):
    q, k, v = rnd(2, Sq, 4, 64, seed=1), rnd(2, Sk, 2, 64, seed=2), rnd(2, Sk, 2, 64, seed=3)
    for causal in (False, True):
        assert_close(hp.prefill("api", mfa, capi, q, k, v, causal), hp.sdpa_gpu(q, k, v, causal), p_rounded=True, what=f"{Sq}x{Sk} causal={causal}")


def test_determinism_and_batch_independence(mfa, capi):
    """reference tests/test_mha.py:93-107 (bitwise repeatability) and :173-192 (batch slices independent)."""
    q, k, v = (rnd(4, 384, 8, 128, seed=s) for s in (1, 2, 3))
    a = hp.prefill("api", mfa, capi, q, k, v, True)
    b = hp.prefill("api", mfa, capi, q, k, v, True)
    assert torch.equal(a, b)
    one = hp.prefill("api", mfa, capi, q[1:2].contiguous(), k[1:2].contiguous(), v[1:2].contiguous(), True)
    assert torch.equal(one[0], a[1])


def test_forced_rescale_and_extreme_scores(mfa, capi, oracle):
    """A key that dominates late forces the online-softmax rescale branch at a chosen tile (guide rule 26);
    large-magnitude scores check the exp2/max path does not overflow."""
    q, k, v = rnd(1, 96, 2, 128, seed=1), rnd(1, 400, 2, 128, seed=2), rnd(1, 400, 2, 128, seed=3)
    k[0, 290, 0] = q[0, 17, 0] * 3
    k[0, 70, 1] = q[0, 80, 1] * 5
    assert_close(hp.prefill("capi", mfa, capi, q, k, v, False), hp.sdpa_gpu(q, k, v, False), p_rounded=True, what="spike")
    q2, k2 = q * 6, k * 6   # scores ~ N(0, 36*sqrt(128)): softmax is nearly one-hot
    out = hp.prefill("capi", mfa, capi, q2, k2, v, False)
    assert_close(out, hp.sdpa_gpu(q2, k2, v, False), atol=4e-3, p_rounded=True, what="large scores")


def test_strided_inputs_and_out_argument(mfa, capi):
    """Packed QKV (B,S,3,H,D) views: row/head/batch strides differ from the contiguous case (api.cpp:58-74)."""
    qkv = rnd(2, 200, 3, 4, 64, seed=9)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    ref = hp.sdpa_gpu(q, k, v, True)
    for route in hp.ROUTES:
        assert_close(hp.prefill(route, mfa, capi, q, k, v, True), ref, p_rounded=True, what=f"strided {route}")
    import mini_flash_attention._C as C
    out = torch.full_like(q.contiguous(), float("nan"))
    ret = C.mini_flash_attention_forward(q, k, v, out, True, -1, -1)
    assert ret.data_ptr() == out.data_ptr()
    assert_close(out, ref, p_rounded=True, what="out=")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_varlen_mixed_lengths(mfa, capi, oracle, dtype):
    """reference tests/test_varlen.py: uniform, mixed, GQA 24:8, short (8..48), 16 random sequences."""
    cases = [([128] * 4, 8, 8, 64), ([128, 256, 512], 8, 8, 64), ([64, 200, 7, 333], 24, 8, 128),
             ([8, 16, 24, 32, 48], 4, 4, 64), (torch.randint(1, 400, (16,), generator=torch.Generator().manual_seed(5)).tolist(), 4, 2, 128)]
    for lens, H, Hk, D in cases:
        tot = sum(lens)
        cu = torch.tensor([0] + lens).cumsum(0).int().to(DEV)
        q, k, v = rnd(tot, H, D, dtype=dtype, seed=1), rnd(tot, Hk, D, dtype=dtype, seed=2), rnd(tot, Hk, D, dtype=dtype, seed=3)
        for causal in (False, True):
            ref = oracle.sdpa_varlen(q.cpu(), k.cpu(), v.cpu(), cu.cpu(), cu.cpu(), causal)
            for route in hp.ROUTES:
                out = hp.prefill(route, mfa, capi, q, k, v, causal, cu_q=cu, cu_k=cu, max_sq=max(lens), max_sk=max(lens))
                assert_close(out, ref, p_rounded=True, what=f"varlen {lens[:4]} {route} causal={causal}")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_varlen_even_batches_take_the_64_row_kernel(mfa, capi, oracle, dtype):
    """Head dim 128, mean length >= 0.9 max: the launcher hands the batch to prefill64_kernel's varlen instances (the route
    query says so); a batch whose work sits in short sequences stays on the general kernel.  Values against per-sequence
    SDPA-fp32."""
    lib = capi.load()
    # (24 heads: 256-row work items for more than half the CUs, below which the launcher keeps the general kernel -- the last case)
    for lens, H, want64 in (([512] * 6, 24, True), ([640, 600, 620, 577, 610], 24, True), ([1024, 1000, 1024], 24, True),
                            ([600] + [300] * 30, 24, False), ([1024, 1000, 1024], 6, False)):
        tot = sum(lens)
        cu = torch.tensor([0] + lens).cumsum(0).int().to(DEV)
        q, k, v = rnd(tot, H, 128, dtype=dtype, seed=1), rnd(tot, H // 3, 128, dtype=dtype, seed=2), rnd(tot, H // 3, 128, dtype=dtype, seed=3)
        for causal in (False, True):
            ref = oracle.sdpa_varlen(q.cpu(), k.cpu(), v.cpu(), cu.cpu(), cu.cpu(), causal)
            for route in hp.ROUTES:
                out = hp.prefill(route, mfa, capi, q, k, v, causal, cu_q=cu, cu_k=cu, max_sq=max(lens), max_sk=max(lens))
                bits = lib.mfa_debug_last_route()
                assert bits & capi.MFA_ROUTE_PREFILL
                if "MFA_PREFILL64" not in os.environ:
                    assert bool(bits & capi.MFA_ROUTE_PREFILL64) == want64, (lens, route, bits)
                assert_close(out, ref, p_rounded=True, what=f"varlen {lens} {route} causal={causal}")


@pytest.mark.parametrize("H,Hk", [(16, 16), (18, 6)])
def test_varlen_ragged_long_batches(mfa, capi, H, Hk):
    """Ragged batches of long sequences, head dim 128 (one long sequence beside short ones; lengths spread over 100 .. 3000; an
    empty sequence): the launcher's length-sorted schedule of the 64-row kernel where long sequences carry the work (the route
    query says which kernel ran), the general kernel otherwise; values against per-sequence fp32 attention on the GPU."""
    lib = capi.load()
    g = torch.Generator().manual_seed(11)
    for lens, want64 in (([4096] + [256] * 9, True), (torch.randint(100, 3001, (12,), generator=g).tolist(), True),
                         ([2000, 0, 1500, 3, 700], True), ([1024] + [200] * 40, False), ([600, 400], False),
                         (torch.randint(300, 1501, (100,), generator=g).tolist(), True), ([1200] + [520] * 199, True)):
        tot = sum(lens)
        cu = torch.tensor([0] + lens).cumsum(0).int().to(DEV)
        q, k, v = rnd(tot, H, 128, dtype=torch.bfloat16, seed=1), rnd(tot, Hk, 128, dtype=torch.bfloat16, seed=2), rnd(tot, Hk, 128, dtype=torch.bfloat16, seed=3)
        for causal in (False, True):
            for route in hp.ROUTES:
                out = hp.prefill(route, mfa, capi, q, k, v, causal, cu_q=cu, cu_k=cu, max_sq=max(lens), max_sk=max(lens))
                if "MFA_PREFILL64" not in os.environ:  # (the launcher's own choice; a forced-route run skips the check)
                    assert bool(lib.mfa_debug_last_route() & capi.MFA_ROUTE_PREFILL64) == want64, (lens[:4], route)
                # (more than 16 sequences: a seeded sample of them -- each also as a batch of one, which the schedule must
                #  reproduce bit for bit)
                picks = range(len(lens)) if len(lens) <= 16 else sorted(torch.randperm(len(lens), generator=g)[:12].tolist())
                for i in picks:
                    n = lens[i]
                    if n == 0:
                        continue
                    s0 = int(cu[i])
                    ref = hp.sdpa_gpu(q[s0:s0 + n][None], k[s0:s0 + n][None], v[s0:s0 + n][None], causal)[0]
                    assert_close(out[s0:s0 + n], ref, p_rounded=True, what=f"ragged {lens[:4]} seq {i} {route} causal={causal}")
                    if len(lens) > 16 and route == "api":
                        c1 = torch.tensor([0, n], dtype=torch.int32, device=DEV)
                        alone = mfa.flash_attn_varlen_func(q[s0:s0 + n], k[s0:s0 + n], v[s0:s0 + n], c1, c1, n, n, causal=causal)
                        lone64 = bool(lib.mfa_debug_last_route() & capi.MFA_ROUTE_PREFILL64)
                        if lone64 == want64:  # (same kernel on both sides: the same arithmetic in the same order)
                            assert torch.equal(alone, out[s0:s0 + n]), f"ragged seq {i}: differs from the batch of one"


def test_varlen_different_q_and_k_lengths(mfa, capi, oracle):
    """total_k != total_q is accepted (the reference insists on equality, api.cpp:259-260)."""
    lq, lk = [3, 70, 128], [200, 64, 129]
    cuq = torch.tensor([0] + lq).cumsum(0).int().to(DEV)
    cuk = torch.tensor([0] + lk).cumsum(0).int().to(DEV)
    q, k, v = rnd(sum(lq), 4, 64, seed=1), rnd(sum(lk), 2, 64, seed=2), rnd(sum(lk), 2, 64, seed=3)
    out = hp.prefill("api", mfa, capi, q, k, v, False, cu_q=cuq, cu_k=cuk, max_sq=max(lq), max_sk=max(lk))
    assert_close(out, oracle.sdpa_varlen(q.cpu(), k.cpu(), v.cpu(), cuq.cpu(), cuk.cpu(), False), p_rounded=True, what="varlen q!=k")


@pytest.mark.parametrize("page", [16, 32, 48, 64, 256])
def test_varlen_paged_prefill_values(mfa, capi, oracle, page):
    """reference tests/test_varlen_block_table.py (pages 16/32/64, scattered pages, mixed Sq): values checked here;
    pages are resolved per key, so page sizes below the 64-key tile and non-powers of two are exact too."""
    lq, lk = [1, 8, 16, 100], [100, 37, 16, 300]
    cuq = torch.tensor([0] + lq).cumsum(0).int().to(DEV)
    cuk = torch.tensor([0] + lk).cumsum(0).int().to(DEV)
    nblk = [(n + page - 1) // page for n in lk]
    pool = sum(nblk) + 5
    perm = torch.randperm(pool, generator=torch.Generator().manual_seed(page))
    table = torch.zeros(len(lk), max(nblk), dtype=torch.int32)
    pos = 0
    for b, n in enumerate(nblk):
        table[b, :n] = perm[pos:pos + n].int()
        pos += n
    table = table.to(DEV)
    q, kp, vp = rnd(sum(lq), 4, 64, seed=1), rnd(pool, page, 2, 64, seed=2), rnd(pool, page, 2, 64, seed=3)
    for causal in (False, True):
        ref = oracle.sdpa_varlen(q.cpu(), kp.cpu(), vp.cpu(), cuq.cpu(), cuk.cpu(), causal, block_table=table.cpu())
        for route in hp.ROUTES:
            out = hp.prefill(route, mfa, capi, q, kp, vp, causal, cu_q=cuq, cu_k=cuk, max_sq=max(lq), max_sk=max(lk), block_table=table)
            assert_close(out, ref, p_rounded=True, what=f"paged prefill page={page} {route} causal={causal}")


def test_baseline_config2_full_size(mfa, capi):
    """BASELINE config 2 (fp16 B48 S1024 H24 D128 causal): checked against SDPA-fp32 on the GPU batch by batch,
    plus size-independent properties: batch-slice independence, invariance to a permutation of batch/head, and
    V-linearity O(V1+V2) = O(V1)+O(V2) to rounding."""
    B, S, H, D = 48, 1024, 24, 128
    q, k, v = (rnd(B, S, H, D, seed=s) for s in (11, 12, 13))
    out = hp.prefill("api", mfa, capi, q, k, v, True)
    for b in (0, 17, 47):
        assert_close(out[b:b + 1], hp.sdpa_gpu(q[b:b + 1], k[b:b + 1], v[b:b + 1], True), p_rounded=True, what=f"config2 batch {b}")
        strict_report(out[b:b + 1], hp.sdpa_gpu(q[b:b + 1], k[b:b + 1], v[b:b + 1], True), f"config2 full size fp16 B48 S1024 H24 D128 causal, batch {b}")
    assert torch.equal(hp.prefill("capi", mfa, capi, q, k, v, True), out)
    pb = torch.randperm(B, device=DEV)
    ph = torch.randperm(H, device=DEV)
    out_p = hp.prefill("api", mfa, capi, q[pb][:, :, ph].contiguous(), k[pb][:, :, ph].contiguous(), v[pb][:, :, ph].contiguous(), True)
    assert torch.equal(out_p, out[pb][:, :, ph])
    v2 = rnd(B, S, H, D, seed=14)
    o1 = hp.prefill("api", mfa, capi, q[:4], k[:4], v[:4], True).float()
    o2 = hp.prefill("api", mfa, capi, q[:4], k[:4], v2[:4], True).float()
    o12 = hp.prefill("api", mfa, capi, q[:4], k[:4], (v[:4].float() + v2[:4].float()).half(), True).float()
    assert (o12 - (o1 + o2)).abs().max() < 8e-3
    # non-causal at the same size, one batch element spot-checked
    outn = hp.prefill("api", mfa, capi, q, k, v, False)
    assert_close(outn[5:6], hp.sdpa_gpu(q[5:6], k[5:6], v[5:6], False), p_rounded=True, what="config2 non-causal")


def test_long_sequences(mfa, capi):
    """S = 16384 and 32768: index arithmetic far past the sizes the reference tests (max 8192, tests/test_gqa.py:334)."""
    for S, causal in ((16384, True), (32768, True), (20000, False)):
        q, k, v = rnd(1, S, 2, 128, dtype=torch.bfloat16, seed=1), rnd(1, S, 1, 128, dtype=torch.bfloat16, seed=2), rnd(1, S, 1, 128, dtype=torch.bfloat16, seed=3)
        out = hp.prefill("api", mfa, capi, q, k, v, causal)
        # reference on a sample of rows (a full fp32 score matrix would be 8.6 GB at S = 32768)
        rows = torch.tensor([0, 1, 63, 64, 4095, S // 2, S - 129, S - 1], device=DEV)
        qs = q[:, rows].float()
        s = torch.einsum("bqhd,bkd->bhqk", qs, k[:, :, 0].float()) / 128 ** 0.5
        if causal:
            s = s.masked_fill(torch.arange(S, device=DEV)[None, None, None, :] > rows[None, None, :, None], float("-inf"))
        ref = torch.einsum("bhqk,bkd->bqhd", torch.softmax(s, -1), v[:, :, 0].float())
        assert_close(out[:, rows], ref, p_rounded=True, what=f"S={S} causal={causal}")
        assert torch.isfinite(out.float()).all()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("per_tile_growth", [1.0, 4.0, 5.9, 6.1, 13.0])
def test_deferred_rescale_slowly_growing_max(mfa, capi, dtype, per_tile_growth):
    """The kernel moves its reference max only when a row's max grew by more than THR = 6 (log2 units) since the last
    move, so P can reach 2^6 in between.  Scores that ramp by a fixed amount per 64-key tile sit just below / above
    the threshold for many consecutive tiles (guide rule 26: the rare branch needs an input that forces it)."""
    S, D = 640, 128
    c = (D ** -0.5) * 1.4426950408889634
    slope = per_tile_growth / (64 * c)            # raw-score growth per key
    q = rnd(1, 96, 2, D, dtype=dtype, seed=1) * 0.05
    k = rnd(1, S, 2, D, dtype=dtype, seed=2) * 0.05
    v = rnd(1, S, 2, D, dtype=dtype, seed=3)
    q[..., 0] = 4.0
    k[..., 0] = (torch.arange(S, device=DEV, dtype=torch.float32) * slope / 4.0).to(dtype)[None, :, None]
    for causal in (False, True):
        out = hp.prefill("capi", mfa, capi, q, k, v, causal)
        assert_close(out, hp.sdpa_gpu(q, k, v, causal), p_rounded=True, what=f"growth={per_tile_growth} causal={causal}")
