"""Perf sweep on the GPU box (BASELINE shapes): prefill fp16 B48 H24 D128 S in {256..4096} causal/non-causal,
decode bf16 B24 24/8 D128 Skv in {512..8192} + the README MHA shape; interleaved rounds in one process, median and
min of per-iteration event timings (reference compare_prefill.py:13-28 method: 5 warm-up + 20 iterations).
  python tools/perf_sweep.py [prefill|decode|all] [--quick]
"""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("MFA_PKG_DIR") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mini-flash-attention_amd"))
import mini_flash_attention as mfa
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _knobs; _knobs.apply()  # noqa: E402


def measure(fn, warmup=5, iters=20):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def prefill(quick):
    B, H, D = 48, 24, 128
    for S in ((1024, 4096) if quick else (256, 512, 1024, 2048, 4096)):
        q, k, v = (torch.randn(B, S, H, D, device="cuda", dtype=torch.float16) for _ in range(3))
        for causal in (True, False):
            med, mn = measure(lambda: mfa.flash_attn_func(q, k, v, causal=causal), iters=10 if S >= 2048 else 20)
            fl = 4.0 * B * H * S * S * D * (0.5 if causal else 1.0)
            by = 8.0 * B * S * H * D
            print(f"prefill fp16 B{B} H{H} D{D} S{S:5d} causal={int(causal)}: med {med:8.3f} ms  min {mn:8.3f} ms  "
                  f"{fl / med / 1e9:7.1f} TFLOP/s ({fl / med / 1e9 / 25:.1f}% MFMA)  {by / med / 1e6:7.1f} GB/s", flush=True)
        del q, k, v
    # GQA + bf16 at S=2048 (guide's comparison shape B16 H64 HKV8)
    q = torch.randn(16, 2048, 64, 128, device="cuda", dtype=torch.bfloat16)
    k, v = (torch.randn(16, 2048, 8, 128, device="cuda", dtype=torch.bfloat16) for _ in range(2))
    for causal in (True, False):
        med, mn = measure(lambda: mfa.flash_attn_func(q, k, v, causal=causal), iters=10)
        fl = 4.0 * 16 * 64 * 2048 * 2048 * 128 * (0.5 if causal else 1.0)
        print(f"prefill bf16 B16 H64/8 D128 S2048 causal={int(causal)}: med {med:8.3f} ms  {fl / med / 1e9:7.1f} TFLOP/s", flush=True)


def decode(quick):
    D = 128
    for (B, H, Hk, dt, name) in ((24, 24, 8, torch.bfloat16, "GQA bf16"), (24, 24, 24, torch.float16, "MHA fp16 (README)")):
        for Sk in ((8192,) if quick else (512, 1024, 2048, 4096, 8192)):
            q = torch.randn(B, 1, H, D, device="cuda", dtype=dt)
            kc, vc = (torch.randn(B, Sk, Hk, D, device="cuda", dtype=dt) for _ in range(2))
            lens = torch.full((B,), Sk, device="cuda", dtype=torch.int32)
            by = 2.0 * (2 * B * Sk * Hk * D + 2 * B * H * D)
            res = []
            for splits in (0, 1, 2, 4, 8, 16):
                med, mn = measure(lambda: mfa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, num_splits=splits))
                res.append(f"s{splits}:{med * 1e3:6.1f}us/{by / med / 1e6:5.0f}GB/s")
            print(f"decode {name} B{B} {H}/{Hk} Skv{Sk:5d}: " + "  ".join(res), flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "all"
    quick = "--quick" in sys.argv
    torch.manual_seed(0)
    if what in ("prefill", "all"):
        prefill(quick)
    if what in ("decode", "all"):
        decode(quick)
