"""Decode split-count sweep for heuristic tuning: small (batch x kv_heads) products on 256 CUs."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mini-flash-attention_amd"))
import mini_flash_attention as mfa
from perf_sweep import measure
torch.manual_seed(0)
D = 128
for (B, H, Hk) in ((1, 24, 8), (2, 24, 8), (4, 24, 8), (8, 24, 8), (16, 24, 8), (1, 32, 32), (4, 8, 1), (12, 24, 8)):
    for Sk in (2048, 8192, 32768):
        q = torch.randn(B, 1, H, D, device="cuda", dtype=torch.bfloat16)
        kc, vc = (torch.randn(B, Sk, Hk, D, device="cuda", dtype=torch.bfloat16) for _ in range(2))
        lens = torch.full((B,), Sk, device="cuda", dtype=torch.int32)
        by = 2.0 * (2 * B * Sk * Hk * D + 2 * B * H * D)
        res = []
        for splits in (0, 1, 2, 4, 8, 16, 32, 64, 128):
            med, mn = measure(lambda: mfa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, num_splits=splits), iters=10)
            res.append(f"s{splits}:{med * 1e3:6.1f}")
        print(f"B{B} {H}/{Hk} Skv{Sk:6d} base={B*Hk:3d} ({by/1e6:7.1f} MB): " + " ".join(res), flush=True)
