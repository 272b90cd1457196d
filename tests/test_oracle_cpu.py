"""CPU-only: pins the C restatement (oracle/mfa_oracle.c) against the golden fixtures and against torch SDPA
fp32 — the oracle of the reference's own tests — at the reference's test shapes (small ones) and thresholds.
Mirrors reference tests/test_mha.py, test_causal.py, test_gqa.py, test_arbitrary_seqlen.py, test_varlen.py,
test_flash_decoding.py in what is covered; nothing here touches the GPU or the HIP library."""
import numpy as np
import pytest
import torch

from conftest import assert_close, from_bits, load_golden


def rnd(*shape, dtype=torch.float16, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g).to(dtype)


# ---- 16-bit conversions the restatement is built on ------------------------------------------------
def test_conversions_match_torch(oracle):
    lib = oracle.lib()
    g = torch.Generator().manual_seed(1)
    vals = torch.cat([torch.randn(2000, generator=g) * 10 ** torch.randint(-8, 6, (2000,), generator=g).float(),
                      torch.tensor([0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e-8, 5.96e-8, 2.98e-8, 6.1e-5, float("inf")])])
    for x in vals.tolist():
        assert lib.mfa_oracle_f32_to_f16(x) == (torch.tensor(x).half().view(torch.int16).item() & 0xFFFF), x
        assert lib.mfa_oracle_f32_to_bf16(x) == (torch.tensor(x).bfloat16().view(torch.int16).item() & 0xFFFF), x
    for h in list(range(0, 0x7C00, 97)) + [1, 2, 0x3FF, 0x400, 0x7BFF, 0x8001, 0xFBFF]:
        assert lib.mfa_oracle_f16_to_f32(h) == torch.tensor([h], dtype=torch.int32).to(torch.int16).view(torch.float16).float().item()


# ---- golden fixtures ---------------------------------------------------------------------------------
def test_golden_g1_config1_plumbing(oracle):
    """BASELINE config 1: fp32 B2 S128 H4 D64 non-causal via SDPA on CPU.  The fixture is SDPA-fp32; the
    restatement runs the same inputs cast to fp16 and to bf16."""
    g = load_golden("g1_fp32_b2_s128_h4_d64")
    q, k, v, expect = (torch.from_numpy(g[n]) for n in ("q", "k", "v", "expect"))
    live = oracle.sdpa_dense(q, k, v, False)
    assert torch.allclose(live, expect, atol=1e-6, rtol=1e-5)
    for dt in (torch.float16, torch.bfloat16):
        qh, kh, vh = q.to(dt), k.to(dt), v.to(dt)
        out = oracle.restated_prefill(qh, kh, vh, False)
        assert_close(out, oracle.sdpa_dense(qh, kh, vh, False), p_rounded=True, what=f"g1 {dt}")


def test_golden_g2_causal_prefill(oracle):
    g = load_golden("g2_fp16_causal_d128")
    for i in range(int(g["n"])):
        q, k, v = (from_bits(g[f"{n}{i}"], torch.float16) for n in "qkv")
        out = oracle.restated_prefill(q, k, v, True)
        assert_close(out, torch.from_numpy(g[f"expect{i}"]), p_rounded=True, what=f"g2[{i}] S={q.size(1)}")


def test_golden_g3_decode(oracle):
    g = load_golden("g3_bf16_decode_gqa")
    q, k, v = (from_bits(g[n], torch.bfloat16) for n in "qkv")
    for i in range(int(g["n"])):
        lens = torch.from_numpy(g[f"lens{i}"])
        for splits in (1, 2, 7):
            out = oracle.restated_decode(q, k, v, lens, num_splits=splits)
            assert_close(out, torch.from_numpy(g[f"expect{i}"]), what=f"g3[{i}] splits={splits}")


def test_golden_g4_config4_varlen(oracle):
    g = load_golden("g4_fp16_varlen_h8_d64")
    q, k, v = (from_bits(g[n], torch.float16) for n in "qkv")
    cu = torch.from_numpy(g["cu"])
    out = oracle.restated_prefill(q, k, v, True, cu_q=cu, cu_k=cu, max_sq=512, max_sk=512)
    assert_close(out, torch.from_numpy(g["expect"]), p_rounded=True, what="g4 varlen")


def test_golden_g5_paged_decode(oracle):
    g = load_golden("g5_bf16_paged_decode")
    for i in range(int(g["n"])):
        q, k, v = (from_bits(g[f"{n}{i}"], torch.bfloat16) for n in "qkv")
        table, lens = torch.from_numpy(g[f"table{i}"]), torch.from_numpy(g[f"lens{i}"])
        for splits in (1, 3):
            out = oracle.restated_decode(q, k, v, lens, block_table=table, num_splits=splits)
            assert_close(out, torch.from_numpy(g[f"expect{i}"]), what=f"g5[{i}] page={k.size(1)} splits={splits}")


# ---- live SDPA comparisons at the reference tests' shapes (kept small for the CPU suite) -------------
@pytest.mark.parametrize("B,S,H,D", [(1, 64, 1, 32), (2, 128, 4, 64), (1, 200, 2, 96), (1, 129, 2, 128), (1, 65, 1, 256)])
@pytest.mark.parametrize("causal", [False, True])
def test_restated_prefill_vs_sdpa(oracle, B, S, H, D, causal):
    """reference tests/test_mha.py:55-91 / test_causal.py:79-143: max < 0.01, mean < 0.001; ours is tighter."""
    q, k, v = (rnd(B, S, H, D, seed=s) for s in (1, 2, 3))
    out = oracle.restated_prefill(q, k, v, causal)
    ref = oracle.sdpa_dense(q, k, v, causal)
    d = (out.float() - ref).abs()
    assert d.max() < 0.01 and d.mean() < 0.001
    assert_close(out, ref, p_rounded=True, what="prefill")


@pytest.mark.parametrize("Hq,Hk", [(8, 2), (8, 1), (6, 2), (4, 4)])
def test_restated_gqa_vs_sdpa(oracle, Hq, Hk):
    """reference tests/test_gqa.py:102-168 (repeat_interleave oracle)."""
    q, k, v = rnd(1, 96, Hq, 64, seed=1), rnd(1, 96, Hk, 64, seed=2), rnd(1, 96, Hk, 64, seed=3)
    assert_close(oracle.restated_prefill(q, k, v, True), oracle.sdpa_dense(q, k, v, True), p_rounded=True, what="gqa")


@pytest.mark.parametrize("S", [1, 7, 63, 65, 100, 127, 129])
def test_restated_arbitrary_seqlen(oracle, S):
    """reference tests/test_arbitrary_seqlen.py:13,77 ragged lengths."""
    q, k, v = (rnd(1, S, 2, 64, seed=s) for s in (4, 5, 6))
    for causal in (False, True):
        assert_close(oracle.restated_prefill(q, k, v, causal), oracle.sdpa_dense(q, k, v, causal), p_rounded=True, what=f"S={S}")


def test_restated_cross_lengths_top_left_causal(oracle):
    """Sq != Sk: the mask is TOP-LEFT aligned (reference prefill.cuh:416-419), like SDPA is_causal."""
    q, k, v = rnd(1, 40, 2, 64, seed=1), rnd(1, 90, 2, 64, seed=2), rnd(1, 90, 2, 64, seed=3)
    assert_close(oracle.restated_prefill(q, k, v, True), oracle.sdpa_dense(q, k, v, True), p_rounded=True, what="Sq<Sk")
    q2 = rnd(1, 90, 2, 64, seed=7)
    assert_close(oracle.restated_prefill(q2, k[:, :40], v[:, :40], True), oracle.sdpa_dense(q2, k[:, :40], v[:, :40], True), p_rounded=True, what="Sq>Sk")


def test_restated_varlen_and_paged_prefill(oracle):
    """reference tests/test_varlen.py (mixed lengths, GQA) and test_varlen_block_table.py (pages 16/32/64,
    scattered pages) — the latter only checks finiteness upstream; values are checked here."""
    lens = [5, 64, 130]
    cu = torch.tensor([0] + lens, dtype=torch.int32).cumsum(0).int()
    q, k, v = rnd(sum(lens), 6, 64, seed=1), rnd(sum(lens), 2, 64, seed=2), rnd(sum(lens), 2, 64, seed=3)
    out = oracle.restated_prefill(q, k, v, True, cu_q=cu, cu_k=cu, max_sq=max(lens), max_sk=max(lens))
    assert_close(out, oracle.sdpa_varlen(q, k, v, cu, cu, True), p_rounded=True, what="varlen")
    for page in (16, 32, 64):
        nblk = [(n + page - 1) // page for n in lens]
        perm = torch.randperm(sum(nblk) + 3, generator=torch.Generator().manual_seed(page))
        table = torch.zeros(len(lens), max(nblk), dtype=torch.int32)
        kp, vp = rnd(sum(nblk) + 3, page, 2, 64, seed=8), rnd(sum(nblk) + 3, page, 2, 64, seed=9)
        pos = 0
        for b, n in enumerate(nblk):
            table[b, :n] = perm[pos:pos + n].int()
            pos += n
        out = oracle.restated_prefill(q, kp, vp, True, cu_q=cu, cu_k=cu, max_sq=max(lens), max_sk=max(lens), block_table=table)
        assert_close(out, oracle.sdpa_varlen(q, kp, vp, cu, cu, True, block_table=table), p_rounded=True, what=f"paged prefill page={page}")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_restated_decode_splits_and_lse(oracle, dtype):
    """reference tests/test_flash_decoding.py: split (num_splits=2) equals no-split; LSE = ln sum exp(scale*s)."""
    q, kc, vc = rnd(2, 1, 8, 64, dtype=dtype, seed=1), rnd(2, 300, 2, 64, dtype=dtype, seed=2), rnd(2, 300, 2, 64, dtype=dtype, seed=3)
    lens = torch.tensor([300, 129], dtype=torch.int32)
    ref = oracle.sdpa_decode(q, kc, vc, lens)
    base, lse, _, _ = oracle.restated_decode(q, kc, vc, lens, num_splits=1, return_partials=True)
    assert_close(base, ref, what="decode")
    for b in range(2):
        s = torch.einsum("hd,khd->hk", q[b, 0].float(), kc[b, :lens[b]].float().repeat_interleave(4, dim=1)) / 8.0
        assert torch.allclose(lse[b], torch.logsumexp(s, dim=-1), atol=1e-4, rtol=1e-5)
    for splits in (2, 3, 5, 64):
        out, lse_s, o_acc, lse_acc = oracle.restated_decode(q, kc, vc, lens, num_splits=splits, return_partials=True)
        assert (out.float() - base.float()).abs().max() <= 2 * (2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11) * 4
        assert torch.allclose(lse_s, lse, atol=1e-5)
        # empty splits keep -inf LSE (reference decode.cuh:533-536)
        ntiles = (129 + 63) // 64
        per = (ntiles + splits - 1) // splits
        for s_ in range(splits):
            if s_ * per >= ntiles:
                assert torch.isinf(lse_acc[s_, 1]).all() and (lse_acc[s_, 1] < 0).all()


def test_restated_edge_cases(oracle):
    q = rnd(1, 1, 2, 64)
    kc, vc = rnd(1, 64, 2, 64, seed=2), rnd(1, 64, 2, 64, seed=3)
    # zero-length cache: output 0, LSE -inf
    out, lse, _, _ = oracle.restated_decode(q, kc, vc, torch.tensor([0], dtype=torch.int32), return_partials=True)
    assert (out == 0).all() and torch.isinf(lse).all()
    # cache_seqlens None = whole cache (superset of the reference, which dereferences NULL)
    assert_close(oracle.restated_decode(q, kc, vc, None), oracle.sdpa_decode(q, kc, vc, None), what="None lens")
    # empty key set in prefill: rows are 0 (reference: l == 0 -> inv = 1 -> O = 0, prefill.cuh:600-612)
    qp = rnd(1, 4, 2, 64)
    out = oracle.restated_prefill(qp, kc[:, :0], vc[:, :0], False)
    assert (out == 0).all()
    # online-softmax rescale forced: one key dominates late in the sequence (guide rule 26)
    q2, k2, v2 = rnd(1, 8, 1, 64, seed=1), rnd(1, 200, 1, 64, seed=2), rnd(1, 200, 1, 64, seed=3)
    k2[0, 150] = q2[0, 3] * 4
    assert_close(oracle.restated_prefill(q2, k2, v2, False), oracle.sdpa_dense(q2, k2, v2, False), p_rounded=True, what="spike")
