"""Paged varlen prefill (bf16 B16 S2048 24/8 causal) by page size, steady timing, with the dense and plain-varlen launches beside
it (developer probe).  python tools/paged_prefill_point.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mini-flash-attention_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mini_flash_attention as mfa
import hip_path as hp
import _knobs
_knobs.apply()  # e.g. MFA_TEST_KNOBS=nw8=1: the general kernel's 8-wave workgroups

def point(name, f, n=30):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.4:
        for _ in range(5): f()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"{name}: {ms:.4f} ms  {fl / ms / 1e9:.0f} TFLOP/s", flush=True)
    return ms

torch.manual_seed(0)
B, S, H, Hk, D = 16, 2048, 24, 8, 128
fl = 4.0 * B * H * S * S * D * 0.5
q = torch.randn(B * S, H, D, device="cuda", dtype=torch.bfloat16)
kd, vd = (torch.randn(B, S, Hk, D, device="cuda", dtype=torch.bfloat16) for _ in range(2))
cu = torch.arange(0, (B + 1) * S, S, device="cuda", dtype=torch.int32)
point("dense (64-row kernel)", lambda: mfa.flash_attn_func(q.view(B, S, H, D), kd, vd, causal=True))
point("varlen              ", lambda: mfa.flash_attn_varlen_func(q, kd.view(B * S, Hk, D), vd.view(B * S, Hk, D), cu, cu, S, S, causal=True))
for page in (16, 64, 128, 256, 512, 1024):
    kp, vp, table = hp.make_paged(kd, vd, page, seed=1, extra_blocks=0)
    point(f"paged, page {page:5d}   ", lambda: mfa.flash_attn_varlen_func(q, kp, vp, cu, cu, S, S, causal=True, block_table=table))
    ident = torch.arange(table.numel(), device="cuda", dtype=torch.int32).view_as(table)
    kp2, vp2 = kd.view(-1, page, Hk, D), vd.view(-1, page, Hk, D)
    point(f"  same, identity table", lambda: mfa.flash_attn_varlen_func(q, kp2, vp2, cu, cu, S, S, causal=True, block_table=ident))
