"""Seeded random shapes through every entry point against the flash_attn-semantics comparator (testsupport/flash_attn,
pinned to the oracle on the CPU).  The hand-picked cases elsewhere probe known boundaries; this sweep is for the
combinations nobody thought of: odd head counts and group sizes, lengths around tile and page edges, windows that
miss everything, empty sequences, forced split counts larger than the tile count, every kv-cache route."""
import os
import random
import sys

import pytest
import torch

import hip_path as hp
from conftest import HALF_ULP, P_ROUND_ATOL, ROOT

sys.path.insert(0, os.path.join(ROOT, "testsupport"))
import flash_attn as fa  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda"


def close(ours, theirs, what, lse=None, lse_ref=None):
    assert torch.isfinite(ours.float()).all(), f"{what}: non-finite output"
    d = (ours.float() - theirs.float()).abs()
    bound = 2e-3 + P_ROUND_ATOL[ours.dtype] + 2 * HALF_ULP[ours.dtype] * theirs.float().abs()
    assert (d <= bound).all(), f"{what}: {(d - bound).max().item():.5f} over the bound (max diff {d.max().item():.5f})"
    if lse is not None:
        fin = torch.isfinite(lse_ref)
        assert torch.equal(torch.isfinite(lse), fin), f"{what}: LSE -inf pattern differs"
        torch.testing.assert_close(lse[fin], lse_ref[fin], atol=3e-3, rtol=1e-4)


def pick_len(rng, cap):
    edges = [0, 1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257]
    return min(cap, rng.choice(edges + [rng.randint(0, cap) for _ in range(6)]))


def heads(rng):
    hk = rng.choice([1, 2, 3, 4, 8])
    g = rng.choice([1, 1, 2, 3, 4, 5, 6, 8, 12, 16])
    return hk * g, hk


@pytest.mark.parametrize("seed", range(60))
def test_fuzz_dense_and_varlen_prefill(mfa, seed):
    rng = random.Random(1000 + seed)
    torch.manual_seed(seed)
    dtype = rng.choice([torch.float16, torch.bfloat16])
    D = rng.choice([32, 64, 96, 128, 128, 160, 256])
    hq, hk = heads(rng)
    causal = rng.random() < 0.5
    window = rng.choice([(-1, -1), (-1, -1), (rng.randint(0, 200), 0), (rng.randint(0, 90), rng.randint(0, 90)), (-1, rng.randint(0, 50))])
    if rng.random() < 0.5:  # dense (Sq == Sk: top-left and bottom-right alignment coincide)
        B, S = rng.randint(1, 3), max(1, pick_len(rng, 700))
        q, k, v = (torch.randn(B, S, h, D, device=DEV).to(dtype) for h in (hq, hk, hk))
        ours, lse = mfa.flash_attn_func(q, k, v, causal=causal, window_size=window, return_softmax_lse=True)
        theirs, lse_ref, _ = fa.flash_attn_func(q, k, v, causal=causal, window_size=window, return_attn_probs=True)
        close(ours, theirs, f"dense seed={seed} B{B} S{S} {hq}/{hk} D{D} causal={causal} w={window}", lse, lse_ref)
    else:  # packed sequences, equal q / k lengths per sequence, some empty
        lens = [pick_len(rng, 400) for _ in range(rng.randint(1, 6))]
        if sum(lens) == 0:
            lens[0] = 5
        lens_k = list(lens)
        if not causal and window == (-1, -1) and rng.random() < 0.5:  # (alignment only matters with a mask)
            lens_k = [pick_len(rng, 400) for _ in lens]
            if sum(lens_k) == 0:
                lens_k[0] = 3
        cu = torch.tensor([0] + lens, device=DEV, dtype=torch.int32).cumsum(0, dtype=torch.int32)
        cuk = torch.tensor([0] + lens_k, device=DEV, dtype=torch.int32).cumsum(0, dtype=torch.int32)
        q = torch.randn(sum(lens), hq, D, device=DEV).to(dtype)
        k, v = (torch.randn(sum(lens_k), hk, D, device=DEV).to(dtype) for _ in range(2))
        ours, lse = mfa.flash_attn_varlen_func(q, k, v, cu, cuk, max(lens), max(lens_k), causal=causal, window_size=window, return_softmax_lse=True)
        theirs, lse_ref, _ = fa.flash_attn_varlen_func(q, k, v, cu, cuk, max(lens), max(lens_k), causal=causal, window_size=window, return_attn_probs=True)
        close(ours, theirs, f"varlen seed={seed} lens={lens}/{lens_k} {hq}/{hk} D{D} causal={causal} w={window}", lse, lse_ref)


@pytest.mark.parametrize("seed", range(100))
def test_fuzz_kvcache(mfa, seed):
    rng = random.Random(2000 + seed)
    torch.manual_seed(seed)
    dtype = rng.choice([torch.float16, torch.bfloat16])
    D = rng.choice([64, 128, 128, 128, 256, 96, 32])
    hq, hk = heads(rng)
    B = rng.randint(1, 5)
    Sq = rng.choice([1, 1, 1, 2, 3, 5, 8, 17, 40, 70, 150])
    Sk = rng.choice([64, 200, 513, 1024, 2100])
    causal = rng.random() < 0.6
    window = rng.choice([(-1, -1), (-1, -1), (-1, -1), (rng.randint(0, 300), 0), (rng.randint(0, 64), rng.randint(0, 64))])
    if D % 32 != 0 or (D not in (64, 128, 256) and (Sq > 1 or window != (-1, -1))):
        window = (-1, -1)  # (head dims without an MFMA instance only serve plain single-token decoding)
        Sq = 1
    splits = rng.choice([0, 0, 1, 2, 3, 5, 16, 64])
    lens = torch.tensor([pick_len(rng, Sk) for _ in range(B)], dtype=torch.int32, device=DEV)
    q = torch.randn(B, Sq, hq, D, device=DEV).to(dtype)
    kc, vc = torch.randn(B, Sk, hk, D, device=DEV).to(dtype), torch.randn(B, Sk, hk, D, device=DEV).to(dtype)
    kw = dict(cache_seqlens=lens, causal=causal, window_size=window)
    what = f"kvcache seed={seed} B{B} Sq{Sq} Sk{Sk} {hq}/{hk} D{D} {dtype} causal={causal} w={window} splits={splits} lens={lens.tolist()}"
    theirs, lse_ref = fa.flash_attn_with_kvcache(q, kc, vc, return_softmax_lse=True, **kw)
    ours, lse = mfa.flash_attn_with_kvcache(q, kc, vc, num_splits=splits, return_softmax_lse=True, **kw)
    close(ours, theirs, what, lse, lse_ref)
    page = rng.choice([16, 32, 48, 64, 128, 256])
    kp, vp, table = hp.make_paged(kc, vc, page, seed=seed)
    ours_p = mfa.flash_attn_with_kvcache(q, kp, vp, block_table=table, num_splits=splits, **kw)
    close(ours_p, theirs, what + f" paged({page})")


@pytest.mark.parametrize("seed", range(24))
def test_fuzz_long_sequence_prefill_d128(mfa, capi, seed):
    """Head dim 128 with sequences long enough for the launcher's choices of round 3 to matter: even and ragged packed batches
    (static, dealt and length-sorted schedules of the 64-row kernel; the general kernel's pair mappings), any head counts
    (multiples of 8 and not), 1 .. 70 sequences (more than 64: a second register of the schedule's tables), optionally over a paged cache, and the same
    shapes as dense batches with few (batch, head) pairs.  Values against the comparator, per sequence."""
    rng = random.Random(3000 + seed)
    torch.manual_seed(seed)
    dtype = rng.choice([torch.float16, torch.bfloat16])
    hk = rng.choice([1, 2, 3, 4, 8])
    hq = hk * rng.choice([1, 2, 3, 4, 8])
    causal = rng.random() < 0.6
    kind = rng.choice(["even", "ragged", "one_long", "many", "dense"])
    if kind == "dense":
        B, S = rng.randint(1, 3), rng.choice([384, 512, 700, 1024, 1500, 2048])
        q, k, v = (torch.randn(B, S, h, 128, device=DEV).to(dtype) for h in (hq, hk, hk))
        ours, lse = mfa.flash_attn_func(q, k, v, causal=causal, return_softmax_lse=True)
        theirs, lse_ref, _ = fa.flash_attn_func(q, k, v, causal=causal, return_attn_probs=True)
        close(ours, theirs, f"dense seed={seed} B{B} S{S} {hq}/{hk} causal={causal} route={capi.load().mfa_debug_last_route()}", lse, lse_ref)
        return
    if kind == "even":
        base = rng.choice([512, 640, 1024, 1536])
        lens = [base - rng.randint(0, base // 12) for _ in range(rng.randint(2, 12))]
    elif kind == "ragged":
        lens = [rng.randint(1, 3000) for _ in range(rng.randint(2, 16))]
    elif kind == "one_long":
        lens = [rng.choice([2048, 3000, 4096])] + [rng.randint(0, 300) for _ in range(rng.randint(1, 20))]
        rng.shuffle(lens)
    else:
        lens = [rng.choice([0, 64, 200, 513, 600, 900]) for _ in range(rng.randint(65, 70))]
        lens[rng.randrange(len(lens))] = 1200
    while sum(lens) > 24000:
        lens.pop()
    cu = torch.tensor([0] + lens, device=DEV, dtype=torch.int32).cumsum(0, dtype=torch.int32)
    q = torch.randn(sum(lens), hq, 128, device=DEV).to(dtype)
    k, v = (torch.randn(sum(lens), hk, 128, device=DEV).to(dtype) for _ in range(2))
    ours, lse = mfa.flash_attn_varlen_func(q, k, v, cu, cu, max(lens), max(lens), causal=causal, return_softmax_lse=True)
    route = capi.load().mfa_debug_last_route()
    theirs, lse_ref, _ = fa.flash_attn_varlen_func(q, k, v, cu, cu, max(lens), max(lens), causal=causal, return_attn_probs=True)
    what = f"varlen seed={seed} {kind} n={len(lens)} max={max(lens)} {hq}/{hk} {dtype} causal={causal} route={route}"
    close(ours, theirs, what, lse, lse_ref)
    if rng.random() < 0.6:  # the same batch over a paged cache (sequence i's keys in the pages of table row i)
        page = rng.choice([16, 64, 128, 256])
        smax = max(lens)
        kd, vd = torch.zeros(len(lens), smax, hk, 128, device=DEV, dtype=dtype), torch.zeros(len(lens), smax, hk, 128, device=DEV, dtype=dtype)
        for i, n in enumerate(lens):
            kd[i, :n], vd[i, :n] = k[int(cu[i]):int(cu[i]) + n], v[int(cu[i]):int(cu[i]) + n]
        kp, vp, table = hp.make_paged(kd, vd, page, seed=seed)
        ours_p = mfa.flash_attn_varlen_func(q, kp, vp, cu, cu, smax, smax, causal=causal, block_table=table)
        close(ours_p, theirs, what + f" paged({page}) route={capi.load().mfa_debug_last_route()}")
