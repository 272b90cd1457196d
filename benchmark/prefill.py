"""Single-shape prefill profile -- counterpart of the reference's benchmark/prefill.py (reference
benchmark/prefill.py:23-79): one launch of each implementation under torch.profiler with the per-kernel device
time table, then the max |difference| between them, asserted.

    python benchmark/prefill.py [--batch 48 --seqlen 4096 --heads 24 --dim 128 --causal --dtype float16]

Defaults are the reference script's shape (fp16 B=48 S=4096 H=24 D=128, non-causal, unit-normalised q/k/v).  The
`flash_attn` column is the real wheel when importable, otherwise the torch-math comparator under testsupport/
(labelled as such); torch SDPA is always shown.  The profiler table names the HIP kernels that ran: a silent
fallback would show up here."""
import argparse

import torch

from harness import MFMA_PEAK_TFLOPS, comparator, dtype_of, prefill_flops

import mini_flash_attention as mfa


def torch_attention(q, k, v, causal):
    o = torch.nn.functional.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), is_causal=causal)
    return o.transpose(1, 2)


def profiled(label, fn):
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA]) as prof:
        out = fn()
        torch.cuda.synchronize()
    print(f"{label} profiling results:")
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=10))
    dev_us = sum(e.self_device_time_total for e in prof.key_averages())
    return out, dev_us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=48)
    ap.add_argument("--seqlen", type=int, default=4096)
    ap.add_argument("--heads", type=int, default=24)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--causal", action="store_true")
    ap.add_argument("--dtype", default="float16", choices=["float16", "bfloat16"])
    a = ap.parse_args()
    fa, fa_label = comparator()
    dt = dtype_of(a.dtype)
    torch.manual_seed(0)
    q, k, v = (torch.nn.functional.normalize(torch.randn(a.batch, a.seqlen, a.heads, a.dim, device="cuda", dtype=dt), dim=-1)
               for _ in range(3))
    impls = [("mini_flash_attention (HIP, gfx950)", lambda: mfa.flash_attn_func(q, k, v, causal=a.causal)),
             ("torch SDPA", lambda: torch_attention(q, k, v, a.causal)),
             (fa_label, lambda: fa.flash_attn_func(q, k, v, causal=a.causal))]
    for _ in range(3):
        for _, fn in impls:
            fn()
    torch.cuda.synchronize()
    outs = {}
    flops = prefill_flops(a.batch, a.heads, a.seqlen, a.seqlen, a.dim, a.causal)
    for label, fn in impls:
        outs[label], us = profiled(label, fn)
        if us > 0:
            print(f"  -> {us / 1e3:.3f} ms device time, {flops / us / 1e6:.1f} TFLOP/s ({flops / us / 1e6 / MFMA_PEAK_TFLOPS:.3f} of the dense MFMA peak)\n")
    (l0, o0), (l1, o1), (l2, o2) = outs.items()
    print("shape:", tuple(o0.shape), tuple(o1.shape), tuple(o2.shape))
    d_fa = (o0.float() - o2.float()).abs().max().item()
    d_t = (o0.float() - o1.float()).abs().max().item()
    d_ref = (o1.float() - o2.float()).abs().max().item()
    print(f"max |mini_flash_attention - {l2}|: {d_fa:.3e}")
    print(f"max |mini_flash_attention - torch SDPA|: {d_t:.3e}")
    print(f"max |torch SDPA - {l2}|: {d_ref:.3e}")
    # unit-normalised inputs: |o| <= 1; the reference's decode script asserts 0.02 on randn inputs (benchmark/decode.py:57)
    assert max(d_fa, d_t) < 2e-3, "outputs differ"


if __name__ == "__main__":
    main()
