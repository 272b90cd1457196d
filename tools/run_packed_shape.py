"""One packed kv-cache shape N times (for rocprofv3):  python tools/run_packed_shape.py B Sq Hq Hk Skv [iters] [splits]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mini-flash-attention_amd"))
import mini_flash_attention as mfa  # noqa: E402

B, Sq, Hq, Hk, Skv = (int(x) for x in sys.argv[1:6])
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 50
splits = int(sys.argv[7]) if len(sys.argv) > 7 else 0
torch.manual_seed(0)
q = torch.randn(B, Sq, Hq, 128, device="cuda", dtype=torch.bfloat16)
caches = [(torch.randn(B, Skv, Hk, 128, device="cuda", dtype=torch.bfloat16), torch.randn(B, Skv, Hk, 128, device="cuda", dtype=torch.bfloat16)) for _ in range(3)]
lens = torch.full((B,), Skv, device="cuda", dtype=torch.int32)
for i in range(iters):
    kc, vc = caches[i % 3]
    mfa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, causal=Sq > 1, num_splits=splits)
torch.cuda.synchronize()
