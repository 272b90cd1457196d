import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mini-flash-attention_amd"))
import mini_flash_attention as mfa
def bench(fn, n=20, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
B, H, D = 48, 24, 128
for S in (1024, 2048, 4096):
    q, k, v = (torch.randn(B, S, H, D, device="cuda", dtype=torch.float16) for _ in range(3))
    for causal in (True, False):
        ms = bench(lambda: mfa.flash_attn_func(q, k, v, causal=causal), n=30 if S < 4096 else 10)
        fl = 4.0 * B * H * S * S * D * (0.5 if causal else 1.0)
        print(f"S{S} causal={int(causal)}: {ms:.3f} ms {fl / ms / 1e9:.0f} TFLOP/s", flush=True)
