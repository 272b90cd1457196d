"""Packed-row kv-cache kernel on paged caches vs the dense cache (developer tool)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mini-flash-attention_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mini_flash_attention as mfa  # noqa: E402
from perf_packed import timed  # noqa: E402


def run(B, Sq, Hq, Hk, Skv, page):
    torch.manual_seed(0)
    D = 128
    q = torch.randn(B, Sq, Hq, D, device="cuda", dtype=torch.bfloat16)
    lens = torch.full((B,), Skv, device="cuda", dtype=torch.int32)
    sets = []
    for i in range(3):
        if page:
            nb = Skv // page
            kp = torch.randn(B * nb, page, Hk, D, device="cuda", dtype=torch.bfloat16)
            vp = torch.randn(B * nb, page, Hk, D, device="cuda", dtype=torch.bfloat16)
            table = torch.randperm(B * nb, device="cuda").int().view(B, nb)
            sets.append((kp, vp, table))
        else:
            sets.append((torch.randn(B, Skv, Hk, D, device="cuda", dtype=torch.bfloat16), torch.randn(B, Skv, Hk, D, device="cuda", dtype=torch.bfloat16), None))
    i = [0]

    def fn():
        kc, vc, t = sets[i[0] % 3]
        i[0] += 1
        mfa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, block_table=t, causal=Sq > 1)
    us = timed(fn)
    byts = 2 * B * Skv * Hk * D * 2
    print(f"B{B} Sq{Sq} {Hq}/{Hk} Skv{Skv} page={page or 'dense':>5}: {us:7.1f} us  {byts / us / 1e3:6.0f} GB/s")


if __name__ == "__main__":
    shapes = ((24, 1, 16, 8, 8192), (24, 1, 32, 8, 8192), (24, 1, 64, 8, 8192), (24, 4, 24, 8, 8192), (24, 1, 24, 8, 8192), (24, 1, 24, 24, 4096), (16, 1, 24, 8, 4096), (64, 1, 32, 8, 1024), (8, 1, 32, 32, 2048))
    for shape in shapes:
        for page in (0, 256, 16):
            run(*shape, page)
