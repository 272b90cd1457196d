"""Flash decoding with few (batch, KV head) rows (bf16 D128, long caches, auto split count): 8 / 9 / 12 / 16 / 17 rows -- row counts
that are no multiple of 8 used to load the XCDs unevenly (developer probe).  python tools/decode_rows_point.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mini-flash-attention_amd"))
import mini_flash_attention as mfa
for B, H, Hk, S in ((8, 4, 1, 32768), (9, 4, 1, 32768), (12, 4, 1, 32768), (16, 4, 1, 32768), (17, 4, 1, 32768), (3, 12, 3, 32768), (5, 8, 2, 16384), (5, 16, 2, 16384)):
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
    print(f"B{B} Hq{H} Hkv{Hk} Skv{S} rows={B*Hk}: {us:8.1f} us  {by / us / 1e3:6.0f} GB/s of {by / 1e6:.0f} MB", flush=True)
    del caches
