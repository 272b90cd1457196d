"""Same FLOPs, fewer K/V bytes: fp16 B48 S1024 H24 D128 causal with 24 / 8 / 1 KV heads (developer probe: how much of config 2's
time follows its HBM / L2 traffic rather than its arithmetic).  python tools/gqa_point.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mini-flash-attention_amd"))
import mini_flash_attention as mfa
B, S, H, D = 48, 1024, 24, 128
for Hk in (24, 8, 1, 24, 8, 1):
    q = torch.randn(B, S, H, D, device="cuda", dtype=torch.float16)
    k, v = (torch.randn(B, S, Hk, D, device="cuda", dtype=torch.float16) for _ in range(2))
    f = lambda: mfa.flash_attn_func(q, k, v, causal=True)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.4:
        for _ in range(10): f()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 40
    by = 2.0 * (2 * B * S * H * D + 2 * B * S * Hk * D)
    print(f"Hkv={Hk:2d}: {ms:.4f} ms  {4.0 * B * H * S * S * D * 0.5 / ms / 1e9:.0f} TFLOP/s  algorithmic bytes {by / 1e6:.0f} MB")
