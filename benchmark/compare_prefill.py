"""Prefill latency of mini_flash_attention.flash_attn_func vs the flash_attn comparator over sequence lengths.

    python benchmark/compare_prefill.py [--causal] [--seqlens 256,512,1024,2048,4096] [--batch-size 48] ...

Same flags and defaults as the reference's benchmark/compare_prefill.py (:89-100); prints TFLOP/s and the share of
the 2.5 PFLOP/s dense MFMA peak beside the milliseconds.
"""
import argparse

import torch

import harness as hs


def main():
    ap = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    ap.add_argument("--seqlens", default="256,512,1024,2048,4096")
    ap.add_argument("--batch-size", type=int, default=48)
    ap.add_argument("--heads", type=int, default=24)
    ap.add_argument("--kv-heads", type=int, default=0, help="0 = same as --heads (MHA)")
    ap.add_argument("--head-dim", type=int, default=128)
    ap.add_argument("--causal", action="store_true")
    ap.add_argument("--dtype", default="float16", choices=["float16", "bfloat16"])
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--output", default="benchmark/flash_attn_seq_len.png", help="chart path ('' = none); a .json goes next to it")
    args = ap.parse_args()

    import mini_flash_attention as mfa
    ref, ref_label = hs.comparator()
    dev, dt = torch.device(args.device), hs.dtype_of(args.dtype)
    hk = args.kv_heads or args.heads
    results, rows = [], []
    with torch.inference_mode():
        for s in hs.parse_int_list(args.seqlens):
            torch.manual_seed(0)
            q = torch.randn(args.batch_size, s, args.heads, args.head_dim, device=dev, dtype=dt)
            k = torch.randn(args.batch_size, s, hk, args.head_dim, device=dev, dtype=dt)
            v = torch.randn(args.batch_size, s, hk, args.head_dim, device=dev, dtype=dt)
            mini = hs.event_timed_ms(lambda: mfa.flash_attn_func(q, k, v, causal=args.causal), args.warmup, args.iters)
            other = hs.event_timed_ms(lambda: ref.flash_attn_func(q, k, v, causal=args.causal), args.warmup, args.iters)
            fl = hs.prefill_flops(args.batch_size, args.heads, s, s, args.head_dim, args.causal)
            tf = fl / (mini["mean"] * 1e-3) / 1e12
            results.append({"x": s, "mini_ms": mini["mean"], "mini_min_ms": mini["min"], "other_ms": other["mean"],
                            "tflops": tf, "mfma_frac": tf / hs.MFMA_PEAK_TFLOPS})
            rows.append([s, f"{mini['mean']:.3f}", f"{other['mean']:.3f}", f"{other['mean'] / mini['mean']:.2f}x",
                         f"{tf:.0f}", f"{100 * tf / hs.MFMA_PEAK_TFLOPS:.1f}%"])
            del q, k, v
    print(f"prefill {args.dtype} B={args.batch_size} H={args.heads}/{hk} D={args.head_dim} causal={args.causal}; comparator: {ref_label}")
    hs.print_table(["seqlen", "mini ms", "other ms", "speedup", "TFLOP/s", "of MFMA peak"], rows)
    hs.save_outputs(results, args.output, "sequence length", "prefill latency", [("mini_ms", "mini-flash-attn (gfx950)"), ("other_ms", ref_label)])


if __name__ == "__main__":
    main()
