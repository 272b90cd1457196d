"""Child process of tests/test_fused_merge_gpu.py: kv-cache attention over a fixed set of seeded shapes, outputs and LSEs
saved to argv[1] (the parent compares two runs made under different launch knobs)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mini-flash-attention_amd"))
import mini_flash_attention as mfa  # noqa: E402

from mini_flash_attention import capi  # noqa: E402  (the same libmfa_hip.so the extension is linked to)

dev = torch.device("cuda", 0)
out = {}
routes = []
g = torch.Generator().manual_seed(7)
# (B, Sq, Hq, Hk, Sk, D, page or 0, splits)
# (cases 2, 4 and 7 have a small (batch, KV head) row count that is no multiple of 8: their splits are spread over all XCDs and
# merged by the combine launch)
cases = [(4, 1, 6, 2, 2500, 128, 0, 5), (2, 1, 8, 8, 4100, 64, 0, 7), (4, 1, 4, 1, 1500, 128, 0, 0), (1, 1, 24, 8, 9000, 128, 0, 0),
         (5, 1, 16, 2, 3000, 128, 0, 6), (4, 3, 16, 2, 2000, 128, 0, 4), (3, 1, 24, 8, 2048, 128, 256, 4), (2, 5, 8, 1, 1300, 64, 64, 3),
         (2, 1, 32, 4, 5000, 256, 0, 9), (6, 2, 12, 4, 777, 96, 0, 2)]
for n, (B, Sq, Hq, Hk, Sk, D, page, splits) in enumerate(cases):
    q = torch.randn(B, Sq, Hq, D, generator=g).to(torch.bfloat16).to(dev)
    kc = torch.randn(B, Sk, Hk, D, generator=g).to(torch.bfloat16).to(dev)
    vc = torch.randn(B, Sk, Hk, D, generator=g).to(torch.bfloat16).to(dev)
    lens = torch.randint(1, Sk + 1, (B,), generator=g).int().to(dev)
    kw = dict(cache_seqlens=lens, num_splits=splits, causal=True)
    if page:
        nb = (Sk + page - 1) // page
        perm = torch.randperm(B * nb, generator=g).to(dev)
        pad = nb * page - Sk
        kp = torch.zeros(B * nb, page, Hk, D, dtype=torch.bfloat16, device=dev)
        vp = torch.zeros_like(kp)
        kp[perm] = torch.nn.functional.pad(kc, (0, 0, 0, 0, 0, pad)).reshape(B * nb, page, Hk, D)
        vp[perm] = torch.nn.functional.pad(vc, (0, 0, 0, 0, 0, pad)).reshape(B * nb, page, Hk, D)
        o = mfa.flash_attn_with_kvcache(q, kp, vp, block_table=perm.int().view(B, nb), **kw)
    else:
        o = mfa.flash_attn_with_kvcache(q, kc, vc, **kw)
    routes.append(capi.load().mfa_debug_last_route())
    for rep in range(3):  # (the arrival counters must be back at zero: repeated launches agree bit for bit)
        o2 = mfa.flash_attn_with_kvcache(q, kp, vp, block_table=perm.int().view(B, nb), **kw) if page else mfa.flash_attn_with_kvcache(q, kc, vc, **kw)
        assert torch.equal(o, o2), f"case {n}: launch {rep + 2} differs from the first"
    out[f"o{n}"] = o.float().cpu()
torch.cuda.synchronize()
torch.save(out, sys.argv[1])
print("saved", len(out))
print("routes", " ".join(str(r) for r in routes))
print("units", " ".join(str(B * Hk) for (B, Sq, Hq, Hk, Sk, D, page, splits) in cases))
