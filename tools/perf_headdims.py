"""Prefill throughput across head dims (developer tool): fp16/bf16, B16, S=2048, H chosen so H*D = 4096."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mini-flash-attention_amd"))
import mini_flash_attention as mfa  # noqa: E402


def timed(fn, warmup=10, iters=40):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


B, S = 16, 2048
for D in (32, 64, 96, 128, 160, 192, 256):
    H = 4096 // D if 4096 % D == 0 else 4096 // D
    for causal in (True, False):
        torch.manual_seed(0)
        q, k, v = (torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
        ms = timed(lambda: mfa.flash_attn_func(q, k, v, causal=causal))
        fl = 4.0 * B * H * S * S * D / (2 if causal else 1)
        print(f"D={D:3d} H={H:3d} causal={int(causal)}: {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s")
