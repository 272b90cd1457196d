"""Timing points of the GENERAL prefill kernel (probe for tools/ab_libs.sh, AB_POINT=tools/short_point.py): short head-dim-128
shapes (fp16 B48 H24), a varlen batch and a head-dim-64 shape; ms per launch after 0.3 s of back-to-back launches."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mini-flash-attention_amd"))
import mini_flash_attention as mfa

def point(name, f, n=60):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        for _ in range(10): f()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}={e0.elapsed_time(e1) / n:.4f}")

B, H, D = 48, 24, 128
for S, causal in ((128, True), (256, True), (256, False), (384, True)):
    q, k, v = (torch.randn(B, S, H, D, device="cuda", dtype=torch.float16) for _ in range(3))
    point(f"S{S}{'c' if causal else 'n'}", lambda: mfa.flash_attn_func(q, k, v, causal=causal))
Bv, Sv = 16, 2048
q = torch.randn(Bv * Sv, 24, 128, device="cuda", dtype=torch.bfloat16)
k, v = (torch.randn(Bv * Sv, 8, 128, device="cuda", dtype=torch.bfloat16) for _ in range(2))
cu = torch.arange(0, (Bv + 1) * Sv, Sv, device="cuda", dtype=torch.int32)
point("varlen2048c", lambda: mfa.flash_attn_varlen_func(q, k, v, cu, cu, Sv, Sv, causal=True), 20)
q, k, v = (torch.randn(16, 2048, 16, 64, device="cuda", dtype=torch.float16) for _ in range(3))
point("D64S2048c", lambda: mfa.flash_attn_func(q, k, v, causal=True), 20)
