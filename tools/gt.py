import os, sys
import torch
sys.path.insert(0, os.environ.get("MFA_PKG_DIR") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mini-flash-attention_amd"))
import mini_flash_attention as mfa
from perf_sweep import measure
torch.manual_seed(0)
for (B, Hq, Hkk) in ((24, 24, 8), (24, 24, 24), (24, 32, 8), (24, 8, 1), (24, 64, 8), (24, 48, 8)):
    qq = torch.randn(B, 1, Hq, 128, device="cuda", dtype=torch.bfloat16)
    kk, vv = (torch.randn(B, 8192, Hkk, 128, device="cuda", dtype=torch.bfloat16) for _ in range(2))
    ll = torch.full((B,), 8192, device="cuda", dtype=torch.int32)
    byy = 2.0 * (2 * B * 8192 * Hkk * 128 + 2 * B * Hq * 128)
    med, mn = measure(lambda: mfa.flash_attn_with_kvcache(qq, kk, vv, cache_seqlens=ll), iters=20)
    print(f"{'PREV' if os.environ.get('MFA_PKG_DIR') else 'CUR '} decode bf16 B{B} Skv8192 {Hq}/{Hkk} D128: med {med*1e3:.1f} us {byy/med/1e6:.0f} GB/s", flush=True)
