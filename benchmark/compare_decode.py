"""Decode (Sq = 1 by default) latency of mini_flash_attention.flash_attn_with_kvcache vs the flash_attn comparator
over KV-cache lengths.

    python benchmark/compare_decode.py [--seqlens 512,1024,2048,4096,8192] [--batch-size 96] [--heads 48] ...

Same flags and defaults as the reference's benchmark/compare_decode.py (:91-102) plus --kv-heads; prints the
K+V streaming rate in GB/s and its share of the 8 TB/s HBM peak beside the milliseconds.
"""
import argparse

import torch

import harness as hs


def main():
    ap = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    ap.add_argument("--seqlens", default="512,1024,2048,4096,8192", help="KV cache lengths")
    ap.add_argument("--seqlen-q", type=int, default=1)
    ap.add_argument("--batch-size", type=int, default=96)
    ap.add_argument("--heads", type=int, default=48)
    ap.add_argument("--kv-heads", type=int, default=0, help="0 = same as --heads (MHA)")
    ap.add_argument("--head-dim", type=int, default=128)
    ap.add_argument("--dtype", default="float16", choices=["float16", "bfloat16"])
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--num-splits", type=int, default=0, help="0 = library heuristic")
    ap.add_argument("--output", default="benchmark/flash_attn_decode.png", help="chart path ('' = none); a .json goes next to it")
    args = ap.parse_args()

    import mini_flash_attention as mfa
    ref, ref_label = hs.comparator()
    dev, dt = torch.device(args.device), hs.dtype_of(args.dtype)
    hk = args.kv_heads or args.heads
    results, rows = [], []
    with torch.inference_mode():
        for skv in hs.parse_int_list(args.seqlens):
            torch.manual_seed(0)
            q = torch.randn(args.batch_size, args.seqlen_q, args.heads, args.head_dim, device=dev, dtype=dt)
            kc = torch.randn(args.batch_size, skv, hk, args.head_dim, device=dev, dtype=dt)
            vc = torch.randn(args.batch_size, skv, hk, args.head_dim, device=dev, dtype=dt)
            lens = torch.full((args.batch_size,), skv, device=dev, dtype=torch.int32)
            kw = {"causal": True} if args.seqlen_q > 1 else {}
            mini = hs.event_timed_ms(lambda: mfa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, num_splits=args.num_splits, **kw),
                                     args.warmup, args.iters)
            other = hs.event_timed_ms(lambda: ref.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, **kw), max(1, args.warmup // 2),
                                      max(2, args.iters // 4))
            gbps = hs.decode_bytes(args.batch_size, args.heads, hk, skv, args.head_dim, q.element_size()) / (mini["mean"] * 1e-3) / 1e9
            results.append({"x": skv, "mini_ms": mini["mean"], "mini_min_ms": mini["min"], "other_ms": other["mean"],
                            "gbps": gbps, "hbm_frac": gbps / hs.HBM_PEAK_GBPS})
            rows.append([skv, f"{mini['mean']:.4f}", f"{other['mean']:.3f}", f"{other['mean'] / mini['mean']:.1f}x",
                         f"{gbps:.0f}", f"{100 * gbps / hs.HBM_PEAK_GBPS:.1f}%"])
            del q, kc, vc
    print(f"decode {args.dtype} B={args.batch_size} Sq={args.seqlen_q} H={args.heads}/{hk} D={args.head_dim}; comparator: {ref_label}")
    hs.print_table(["kv_len", "mini ms", "other ms", "speedup", "GB/s", "of HBM peak"], rows)
    hs.save_outputs(results, args.output, "KV cache length", "decode latency", [("mini_ms", "mini-flash-attn (gfx950)"), ("other_ms", ref_label)])


if __name__ == "__main__":
    main()
