"""Single-shape flash-decoding profile -- counterpart of the reference's benchmark/decode.py (reference
benchmark/decode.py:6-58): one launch of each implementation under torch.profiler with the per-kernel device time
table, output shape / dtype / device checks, then max and mean |difference|, asserted with the reference's bounds.

    python benchmark/decode.py [--batch 96 --seqlen-kv 4096 --heads 48 --kv-heads 48 --dim 128 --dtype float16]

Defaults are the reference script's shape (fp16 MHA B=96 Sq=1 Skv=4096 H=48 D=128)."""
import argparse

import torch

from harness import HBM_PEAK_GBPS, comparator, decode_bytes, dtype_of

import mini_flash_attention as mfa


def profiled(label, fn):
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA]) as prof:
        out = fn()
        torch.cuda.synchronize()
    print(f"{label} profiling results:")
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=10))
    return out, sum(e.self_device_time_total for e in prof.key_averages())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=96)
    ap.add_argument("--seqlen-kv", type=int, default=4096)
    ap.add_argument("--heads", type=int, default=48)
    ap.add_argument("--kv-heads", type=int, default=0, help="0 = heads (MHA)")
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--dtype", default="float16", choices=["float16", "bfloat16"])
    a = ap.parse_args()
    hk = a.kv_heads or a.heads
    fa, fa_label = comparator()
    dt = dtype_of(a.dtype)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    q = torch.randn(a.batch, 1, a.heads, a.dim, device=dev, dtype=dt)
    kc, vc = (torch.randn(a.batch, a.seqlen_kv, hk, a.dim, device=dev, dtype=dt) for _ in range(2))
    lens = torch.full((a.batch,), a.seqlen_kv, dtype=torch.int32, device=dev)
    ours = lambda: mfa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens)
    theirs = lambda: fa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens)
    for _ in range(3):
        ours()
        theirs()
    torch.cuda.synchronize()
    out, us = profiled("mini_flash_attention with KV cache (HIP, gfx950)", ours)
    by = decode_bytes(a.batch, a.heads, hk, a.seqlen_kv, a.dim)
    if us > 0:
        print(f"  -> {us:.1f} us device time, {by / us / 1e3:.0f} GB/s ({by / us / 1e3 / HBM_PEAK_GBPS:.3f} of the HBM peak; "
              f"{by / 1e6:.0f} MB of K/V + q + o, cache {'larger' if by > 256e6 else 'smaller'} than the 256 MB Infinity Cache)\n")
    assert out.shape == q.shape, f"output shape {tuple(out.shape)} != input shape {tuple(q.shape)}"
    assert out.dtype == dt and out.device == q.device
    ref, _ = profiled(fa_label + " with KV cache", theirs)
    max_diff = (out.float() - ref.float()).abs().max().item()
    mean_diff = (out.float() - ref.float()).abs().mean().item()
    print(f"max difference between mini_flash_attention and {fa_label}: {max_diff:.6f}")
    print(f"mean difference: {mean_diff:.6f}")
    assert max_diff < 0.02, f"max difference too large: {max_diff:.6f}"
    assert mean_diff < 0.002, f"mean difference too large: {mean_diff:.6f}"


if __name__ == "__main__":
    main()
