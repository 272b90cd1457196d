"""One timing point per headline prefill shape (fp16 B48 H24 D128; ms, steady: 0.4 s of back-to-back launches first) -- the probe
tools/ab_libs.sh runs under each library."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mini-flash-attention_amd"))
import mini_flash_attention as mfa
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _knobs; _knobs.apply()
B, H, D = 48, 24, 128
for S, causal in ((1024, True), (1024, False), (512, True), (256, True), (2048, True), (4096, False)):
    q, k, v = (torch.randn(B, S, H, D, device="cuda", dtype=torch.float16) for _ in range(3))
    f = lambda: mfa.flash_attn_func(q, k, v, causal=causal)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.4:
        for _ in range(10): f()
        torch.cuda.synchronize()
    n = 40 if S <= 1024 else 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    print(f"S{S}{'c' if causal else 'n'}={e0.elapsed_time(e1) / n:.4f}")
