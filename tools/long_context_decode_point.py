"""Flash decoding for few sequences with a long context (bf16 D128, full-length caches): us and GB/s of K+V, auto split count
(developer probe).  python tools/long_context_decode_point.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mini-flash-attention_amd"))
import mini_flash_attention as mfa
for B, H, Hk, S in ((1, 32, 8, 8192), (1, 32, 8, 32768), (1, 32, 8, 131072), (1, 8, 8, 32768), (1, 64, 8, 32768), (4, 32, 8, 32768), (1, 32, 32, 32768), (1, 128, 8, 32768), (2, 8, 1, 65536)):
    by = 2 * B * S * Hk * 128 * 2
    copies = max(2, min(8, int(600e6 / by)))
    q = torch.randn(B, 1, H, 128, device="cuda", dtype=torch.bfloat16)
    caches = [(torch.randn(B, S, Hk, 128, device="cuda", dtype=torch.bfloat16), torch.randn(B, S, Hk, 128, device="cuda", dtype=torch.bfloat16)) for _ in range(copies)]
    cl = torch.full((B,), S, device="cuda", dtype=torch.int32)
    it = [0]
    def f():
        k, v = caches[it[0] % copies]; it[0] += 1
        return mfa.flash_attn_with_kvcache(q, k, v, cache_seqlens=cl)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        for _ in range(10): f()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print(f"B{B} Hq{H} Hkv{Hk} Skv{S}: {us:8.1f} us  {by / us / 1e3:6.0f} GB/s of {by / 1e6:.0f} MB ({copies} rotating caches)", flush=True)
    del caches
