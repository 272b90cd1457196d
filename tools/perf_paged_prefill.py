import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mini-flash-attention_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mini_flash_attention as mfa
from perf_sweep import measure
import hip_path as hp
torch.manual_seed(0)
B, S, H, Hk, D = 16, 2048, 24, 8, 128
q = torch.randn(B * S, H, D, device="cuda", dtype=torch.bfloat16)
kd, vd = (torch.randn(B, S, Hk, D, device="cuda", dtype=torch.bfloat16) for _ in range(2))
cu = torch.arange(0, (B + 1) * S, S, device="cuda", dtype=torch.int32)
fl = 4.0 * B * H * S * S * D * 0.5
dense, _ = measure(lambda: mfa.flash_attn_func(q.view(B, S, H, D), kd, vd, causal=True), iters=10)
print(f"dense (64-rows-per-wave kernel) bf16 B{B} S{S} {H}/{Hk} causal: {dense:.3f} ms {fl/dense/1e9:.1f} TFLOP/s")
med, _ = measure(lambda: mfa.flash_attn_varlen_func(q, kd.view(B * S, Hk, D), vd.view(B * S, Hk, D), cu, cu, S, S, causal=True), iters=10)
print(f"varlen dense  bf16 B{B} S{S} {H}/{Hk} causal: {med:.3f} ms {fl/med/1e9:.1f} TFLOP/s = {dense/med:.2f} x the dense launch")
for page in (256, 64, 16):
    kp, vp, table = hp.make_paged(kd, vd, page, seed=1, extra_blocks=0)
    med, _ = measure(lambda: mfa.flash_attn_varlen_func(q, kp, vp, cu, cu, S, S, causal=True, block_table=table), iters=10)
    print(f"varlen paged{page:4d} bf16 B{B} S{S} {H}/{Hk} causal: {med:.3f} ms {fl/med/1e9:.1f} TFLOP/s = {dense/med:.2f} x the dense launch")
