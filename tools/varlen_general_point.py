"""Packed varlen prefill on the general kernel's head dims (bf16, D 64 / 256, causal): a ragged batch against an even batch of the
same token count, TFLOP/s of the visible scores (developer probe).  python tools/varlen_general_point.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mini-flash-attention_amd"))
import mini_flash_attention as mfa
g = torch.Generator().manual_seed(1)
ragged = {"128..4096 log x 32": (128 * 2 ** (5 * torch.rand(32, generator=g))).int().tolist(), "one 8192 + 31 x 256": [8192] + [256] * 31,
          "512..4096 x 16": torch.randint(512, 4097, (16,), generator=g).tolist()}
for D, H, Hk in ((64, 16, 16), (64, 32, 8), (256, 8, 2)):
    for name, lens in ragged.items():
        for label, ls in ((name, lens), ("  even batch, same tokens", [sum(lens) // len(lens)] * len(lens))):
            T = sum(ls)
            q = torch.randn(T, H, D, device="cuda", dtype=torch.bfloat16)
            k, v = (torch.randn(T, Hk, D, device="cuda", dtype=torch.bfloat16) for _ in range(2))
            cu = torch.tensor([0] + ls, device="cuda").cumsum(0).int()
            f = lambda: mfa.flash_attn_varlen_func(q, k, v, cu, cu, max(ls), max(ls), causal=True)
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.3:
                for _ in range(5): f()
                torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): f()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            fl = sum(4.0 * H * n * n * D * 0.5 for n in ls)
            print(f"D{D} H{H}/{Hk} {label}: {ms:.4f} ms  {fl / ms / 1e9:6.0f} TFLOP/s", flush=True)
