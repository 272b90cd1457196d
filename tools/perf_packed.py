"""kv-cache attention timings for the packed-row kernel vs the other routes (developer tool).
   python tools/perf_packed.py            # library's own routing and split choice
   MFA_KVCACHE_PACKED=0 python tools/perf_packed.py   # force the vector decode / per-head prefill routes
   python tools/perf_packed.py sweep      # forced split counts for a few shapes
GB/s = (K + V of the valid cache + q + o) / time."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mini-flash-attention_amd"))
import mini_flash_attention as mfa
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _knobs; _knobs.apply()  # noqa: E402


def timed(fn, warmup=5, iters=30):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def case(B, Sq, Hq, Hk, Skv, D=128, dtype=torch.bfloat16, splits=0, nbuf=3):
    torch.manual_seed(0)
    q = torch.randn(B, Sq, Hq, D, device="cuda", dtype=dtype)
    caches = [(torch.randn(B, Skv, Hk, D, device="cuda", dtype=dtype), torch.randn(B, Skv, Hk, D, device="cuda", dtype=dtype)) for _ in range(nbuf)]
    lens = torch.full((B,), Skv, device="cuda", dtype=torch.int32)
    i = [0]

    def fn():
        kc, vc = caches[i[0] % nbuf]
        i[0] += 1
        mfa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, causal=Sq > 1, num_splits=splits)
    us = timed(fn)
    byts = 2 * B * Skv * Hk * D * 2 + 2 * B * Sq * Hq * D * 2
    return us, byts / us / 1e3


if __name__ != "__main__":
    pass
elif len(sys.argv) > 1 and sys.argv[1] == "sweep":
    for (B, Sq, Hq, Hk, Skv) in ((24, 1, 64, 8, 8192), (24, 8, 24, 8, 8192), (8, 4, 32, 8, 32768), (64, 1, 32, 4, 2048)):
        row = []
        for s in (0, 1, 2, 3, 4, 6, 8, 12, 16, 32):
            us, gb = case(B, Sq, Hq, Hk, Skv, splits=s)
            row.append(f"s{s}:{us:7.1f}us/{gb:5.0f}")
        print(f"B{B} Sq{Sq} {Hq}/{Hk} Skv{Skv}: " + "  ".join(row))
else:
    print("route override MFA_KVCACHE_PACKED =", os.environ.get("MFA_KVCACHE_PACKED", "(library default)"))
    for (B, Sq, Hq, Hk, Skv) in ((24, 1, 24, 8, 8192), (24, 1, 48, 8, 8192), (24, 1, 64, 8, 8192), (24, 1, 32, 4, 8192), (64, 1, 64, 8, 2048),
                                 (24, 2, 24, 8, 8192), (24, 4, 24, 8, 8192), (24, 8, 24, 8, 8192), (24, 16, 24, 8, 8192), (24, 64, 24, 8, 8192),
                                 (24, 4, 64, 8, 8192), (24, 16, 64, 8, 8192), (4, 8, 32, 8, 65536), (128, 2, 32, 8, 1024)):
        us, gb = case(B, Sq, Hq, Hk, Skv)
        print(f"B{B:3d} Sq{Sq:3d} {Hq}/{Hk} Skv{Skv:6d}: {us:8.1f} us  {gb:6.0f} GB/s ({gb / 80:.0f} % of 8 TB/s)")
