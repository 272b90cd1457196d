"""Flash decoding over a cache with ragged per-sequence lengths (bf16 Hq24 Hkv8 D128, Skv capacity 8192): effective GB/s on
the bytes the lengths imply, against an even batch of the same total length (developer probe).  python tools/ragged_decode_point.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mini-flash-attention_amd"))
import mini_flash_attention as mfa

def point(name, lens, cap=8192, H=24, Hk=8, D=128, copies=3):
    B = len(lens)
    q = torch.randn(B, 1, H, D, device="cuda", dtype=torch.bfloat16)
    caches = [(torch.randn(B, cap, Hk, D, device="cuda", dtype=torch.bfloat16), torch.randn(B, cap, Hk, D, device="cuda", dtype=torch.bfloat16)) for _ in range(copies)]
    cl = torch.tensor(lens, device="cuda", dtype=torch.int32)
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
    for _ in range(60): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 60 * 1e3
    by = 2 * sum(lens) * Hk * D * 2 + 2 * B * H * D * 2
    print(f"{name}: {us:7.1f} us  {by / us / 1e3:6.0f} GB/s of {by / 1e6:.0f} MB", flush=True)

g = torch.Generator().manual_seed(2)
r = torch.randint(256, 8193, (64,), generator=g).tolist()
point("even   64 x mean(random)", [sum(r) // 64] * 64)
point("random 64 x 256..8192   ", r)
point("one 8192 + 63 x 512      ", [8192] + [512] * 63)
point("even   64 x 632          ", [632] * 64)
r2 = torch.randint(256, 8193, (16,), generator=g).tolist()
point("even   16 x mean(random)", [sum(r2) // 16] * 16)
point("random 16 x 256..8192   ", r2)
