"""GPU runtime-contract tests of the boundary (SURVEY.md §8b "Threading / async"): launches go to the CURRENT stream,
never synchronise, and are hipGraph-capturable; several devices' worth of state is not shared (thread-local errors)."""
import threading

import pytest
import torch

import hip_path as hp
from conftest import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rnd(*shape, dtype=torch.float16, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randn(*shape, generator=g).to(dtype).to(DEV)


def test_side_stream_ordering(mfa):
    """The kernels run on the stream that is current at call time (reference api.cpp:182,263,442)."""
    q, k, v = (rnd(2, 512, 8, 128, seed=s) for s in (1, 2, 3))
    ref = mfa.flash_attn_func(q, k, v, causal=True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        q2 = q * 1.0                      # produced on the side stream
        out = mfa.flash_attn_func(q2, k, v, causal=True)
        done = torch.cuda.Event()
        done.record(side)
    done.synchronize()
    assert torch.equal(out, ref)


def test_hip_graph_capture_and_replay(mfa):
    """Prefill, split decode (+ combine), append and the packed-row kv-cache kernel (3 query tokens, split) captured
    into one graph; replay with new inputs in the same buffers."""
    B, S, H, Hk, D = 2, 256, 8, 2, 128
    q = rnd(B, S, H, D, seed=1)
    k, v = rnd(B, S, Hk, D, seed=2), rnd(B, S, Hk, D, seed=3)
    qd = rnd(B, 1, H, D, dtype=torch.bfloat16, seed=4)
    kc, vc = rnd(B, 2048, Hk, D, dtype=torch.bfloat16, seed=5), rnd(B, 2048, Hk, D, dtype=torch.bfloat16, seed=6)
    kn, vn = rnd(B, 1, Hk, D, dtype=torch.bfloat16, seed=7), rnd(B, 1, Hk, D, dtype=torch.bfloat16, seed=8)
    lens = torch.tensor([2000, 777], dtype=torch.int32, device=DEV)
    qs = rnd(B, 3, H, D, dtype=torch.bfloat16, seed=9)
    # warm-up outside capture (lazy module/attribute initialisation)
    mfa.flash_attn_func(q, k, v, causal=True)
    mfa.flash_attn_with_kvcache(qd, kc, vc, cache_seqlens=lens, num_splits=4, k=kn, v=vn)
    mfa.flash_attn_with_kvcache(qs, kc, vc, cache_seqlens=lens, causal=True, num_splits=3)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        o1 = mfa.flash_attn_func(q, k, v, causal=True)
        o2 = mfa.flash_attn_with_kvcache(qd, kc, vc, cache_seqlens=lens, num_splits=4, k=kn, v=vn)
        o3 = mfa.flash_attn_with_kvcache(qs, kc, vc, cache_seqlens=lens, causal=True, num_splits=3)
    for seed in (11, 12):
        q.copy_(rnd(B, S, H, D, seed=seed))
        qd.copy_(rnd(B, 1, H, D, dtype=torch.bfloat16, seed=seed + 100))
        g.replay()
        torch.cuda.synchronize()
        assert_close(o1, hp.sdpa_gpu(q, k, v, True), what="graph prefill")
        qs.copy_(rnd(B, 3, H, D, dtype=torch.bfloat16, seed=seed + 200))
        g.replay()
        torch.cuda.synchronize()
        eager = mfa.flash_attn_with_kvcache(qd, kc, vc, cache_seqlens=lens, num_splits=4, k=kn, v=vn)
        assert torch.equal(o2, eager)
        assert torch.equal(o3, mfa.flash_attn_with_kvcache(qs, kc, vc, cache_seqlens=lens, causal=True, num_splits=3))


def test_thread_local_errors_and_concurrent_calls(mfa, capi):
    """No global mutable state: concurrent callers get their own error strings and correct results."""
    q, k, v = (rnd(1, 128, 4, 64, seed=s) for s in (1, 2, 3))
    ref = mfa.flash_attn_func(q, k, v)
    errs, outs = [], []

    def bad():
        try:
            mfa.flash_attn_func(q[..., :60], k[..., :60], v[..., :60])   # head_dim 60: unsupported
        except RuntimeError as e:
            errs.append(str(e))

    def good():
        outs.append(mfa.flash_attn_func(q, k, v))

    ts = [threading.Thread(target=f) for f in (bad, good, bad, good)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    torch.cuda.synchronize()
    assert len(errs) == 2 and all("head_dim" in e or "contiguous" in e or "multiple" in e for e in errs)
    assert all(torch.equal(o, ref) for o in outs)


def test_graph_capture_with_and_without_split_counters(mfa, capi):
    """The in-kernel split merge uses arrival counters that the HOST layer owns (include/mfa.h: no launch entry point
    allocates).  (a) On a stream that has no counter buffer yet, a capture must not create one: the captured launch
    merges through decode_combine_kernel.  (b) On a stream whose buffer exists, the captured launch merges in the kernel
    and keeps a pointer that stays valid: after unrelated, larger split launches the replay still equals the eager call."""
    lib = capi.load()
    B, H, Hk, D = 4, 8, 2, 128  # (8 (batch, KV head) rows: with fewer the splits go out over all XCDs and never merge in the kernel)
    qd = rnd(B, 1, H, D, dtype=torch.bfloat16, seed=4)
    kc, vc = rnd(B, 2048, Hk, D, dtype=torch.bfloat16, seed=5), rnd(B, 2048, Hk, D, dtype=torch.bfloat16, seed=6)
    lens = torch.tensor([2000, 777, 1500, 64], dtype=torch.int32, device=DEV)
    call = lambda: mfa.flash_attn_with_kvcache(qd, kc, vc, cache_seqlens=lens, num_splits=4)
    eager = call()
    assert lib.mfa_debug_last_route() == capi.MFA_ROUTE_DECODE | capi.MFA_ROUTE_FUSED_MERGE
    torch.cuda.synchronize()
    fresh, warm = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(warm):
        call()
    torch.cuda.synchronize()
    graphs = []
    for stream, want in ((fresh, capi.MFA_ROUTE_COMBINE_LAUNCH), (warm, capi.MFA_ROUTE_FUSED_MERGE)):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            o = call()
            route = lib.mfa_debug_last_route()
        assert route == capi.MFA_ROUTE_DECODE | want, (route, want)
        graphs.append((g, o))
    # other split launches in between (more rows, more splits: a library-owned buffer would have had to grow here)
    qb = rnd(16, 1, 32, D, dtype=torch.bfloat16, seed=7)
    kb, vb = rnd(16, 1024, 8, D, dtype=torch.bfloat16, seed=8), rnd(16, 1024, 8, D, dtype=torch.bfloat16, seed=9)
    with torch.cuda.stream(warm):
        mfa.flash_attn_with_kvcache(qb, kb, vb, num_splits=4)
    torch.cuda.synchronize()
    for seed in (21, 22):
        qd.copy_(rnd(B, 1, H, D, dtype=torch.bfloat16, seed=seed))
        for g, o in graphs:
            g.replay()
        torch.cuda.synchronize()
        eager = call()
        for g, o in graphs:
            assert torch.equal(o, eager)
