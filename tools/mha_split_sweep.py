"""README MHA decode shapes (fp16 B24 H24 D128 Sq1) over rotating caches: forced split counts against auto (developer aid).
  python tools/mha_split_sweep.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mini-flash-attention_amd"))
import mini_flash_attention as mfa
for (B, H, Hk, Sk, dt) in ((24, 24, 24, 512, torch.float16), (24, 24, 24, 1024, torch.float16), (24, 24, 24, 2048, torch.float16), (24, 24, 8, 2048, torch.bfloat16), (16, 24, 8, 4096, torch.bfloat16)):
    by = 2.0 * (2 * B * Sk * Hk * 128 + 2 * B * H * 128)
    copies = max(2, min(8, int(1.5e9 // by)))
    sets = [tuple(torch.randn(B, Sk, Hk, 128, device="cuda", dtype=dt) for _ in range(2)) for _ in range(copies)]
    q = torch.randn(B, 1, H, 128, device="cuda", dtype=dt)
    lens = torch.full((B,), Sk, device="cuda", dtype=torch.int32)
    st = {"i": 0}
    res = []
    for s in (0, 1, 2, 3, 4, 6, 8):
        def run():
            kc, vc = sets[st["i"] % copies]; st["i"] += 1
            mfa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, num_splits=s)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.15:
            for _ in range(20): run()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 200 * 1e3
        res.append(f"s{s}:{us:6.1f}us/{by / us / 1e3:5.0f}GB/s")
    print(f"B{B} {H}/{Hk} Skv{Sk:5d} {str(dt)[6:]}: " + "  ".join(res), flush=True)
    del sets
