"""Chunked prefill over a kv-cache (flash_attn_with_kvcache with seqlen_q > 1, causal = aligned to the last key; bf16 D128):
us and TFLOP/s of the visible scores (developer probe).  python tools/chunked_prefill_point.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mini-flash-attention_amd"))
import mini_flash_attention as mfa
from mini_flash_attention import capi
lib = capi.load()
for B, H, Hk, Sq, Skv in ((1, 32, 8, 1024, 8192), (1, 32, 8, 2048, 8192), (1, 32, 8, 2048, 32768), (2, 32, 8, 1024, 16384), (1, 28, 4, 2048, 16384),
                          (4, 32, 8, 512, 8192), (1, 8, 1, 2048, 32768), (8, 32, 8, 256, 4096), (1, 32, 8, 8192, 8192)):
    q = torch.randn(B, Sq, H, 128, device="cuda", dtype=torch.bfloat16)
    k, v = (torch.randn(B, Skv, Hk, 128, device="cuda", dtype=torch.bfloat16) for _ in range(2))
    cl = torch.full((B,), Skv, device="cuda", dtype=torch.int32)
    f = lambda: mfa.flash_attn_with_kvcache(q, k, v, cache_seqlens=cl, causal=True)
    f(); route = lib.mfa_debug_last_route()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        for _ in range(5): f()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    fl = 4.0 * B * H * 128 * (Sq * (Skv - Sq) + Sq * (Sq + 1) / 2)
    print(f"B{B} Hq{H} Hkv{Hk} Sq{Sq} Skv{Skv}: {us:8.1f} us  {fl / us / 1e6:6.0f} TFLOP/s  route bits {route}", flush=True)
