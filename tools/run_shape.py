"""Run one shape N times (for rocprofv3 passes):  python tools/run_shape.py prefill S causal [iters] | decode Sk splits [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mini-flash-attention_amd"))
import mini_flash_attention as mfa  # noqa: E402

torch.manual_seed(0)
kind = sys.argv[1]
if kind == "prefill":
    S, causal = int(sys.argv[2]), bool(int(sys.argv[3]))
    iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
    B, H, D = 48, 24, 128
    q, k, v = (torch.randn(B, S, H, D, device="cuda", dtype=torch.float16) for _ in range(3))
    for _ in range(iters):
        mfa.flash_attn_func(q, k, v, causal=causal)
else:
    Sk, splits = int(sys.argv[2]), int(sys.argv[3])
    iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
    B, H, Hk, D = 24, 24, 8, 128
    q = torch.randn(B, 1, H, D, device="cuda", dtype=torch.bfloat16)
    kc, vc = (torch.randn(B, Sk, Hk, D, device="cuda", dtype=torch.bfloat16) for _ in range(2))
    lens = torch.full((B,), Sk, device="cuda", dtype=torch.int32)
    for _ in range(iters):
        mfa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, num_splits=splits)
torch.cuda.synchronize()
