"""Flash decoding over a cache with ragged per-sequence lengths (bf16 Hq24 Hkv8 D128, Skv capacity 8192): effective GB/s on
the bytes the lengths imply, against an even batch of the same total length (developer probe).  python tools/ragged_decode_point.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mini-flash-attention_amd"))
import mini_flash_attention as mfa

def point(name, lens, cap=8192, H=24, Hk=8, D=128, copies=3, splits=0):
    B = len(lens)
    q = torch.randn(B, 1, H, D, device="cuda", dtype=torch.bfloat16)
    caches = [(torch.randn(B, cap, Hk, D, device="cuda", dtype=torch.bfloat16), torch.randn(B, cap, Hk, D, device="cuda", dtype=torch.bfloat16)) for _ in range(copies)]
    cl = torch.tensor(lens, device="cuda", dtype=torch.int32)
    it = [0]
    def f():
        k, v = caches[it[0] % copies]; it[0] += 1
        return mfa.flash_attn_with_kvcache(q, k, v, cache_seqlens=cl, num_splits=splits)
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
    print(f"{name} splits={splits}: {us:7.1f} us  {by / us / 1e3:6.0f} GB/s of {by / 1e6:.0f} MB", flush=True)

if len(sys.argv) > 1 and sys.argv[1] == "splits":  # forced split counts on even-full and ragged batches
    g = torch.Generator().manual_seed(2)
    r = torch.randint(256, 8193, (64,), generator=g).tolist()
    r24 = torch.randint(256, 8193, (24,), generator=g).tolist()
    for name, lens in (("full   64 x 8192", [8192] * 64), ("random 64 x 256..8192", r), ("one 8192 + 63 x 512", [8192] + [512] * 63),
                       ("full   24 x 8192 (config 3)", [8192] * 24), ("random 24 x 256..8192", r24), ("one 8192 + 23 x 512", [8192] + [512] * 23)):
        for sp in (0, 1, 2, 4, 8):
            point(name, lens, splits=sp, copies=2 if len(lens) > 24 else 3)
    sys.exit(0)
g = torch.Generator().manual_seed(2)
r = torch.randint(256, 8193, (64,), generator=g).tolist()
point("even   64 x mean(random)", [sum(r) // 64] * 64)
point("random 64 x 256..8192   ", r)
point("one 8192 + 63 x 512      ", [8192] + [512] * 63)
point("even   64 x 632          ", [632] * 64)
r2 = torch.randint(256, 8193, (16,), generator=g).tolist()
point("even   16 x mean(random)", [sum(r2) // 16] * 16)
point("random 16 x 256..8192   ", r2)
