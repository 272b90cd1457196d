"""Shared pieces of the latency comparisons (benchmark/compare_prefill.py, benchmark/compare_decode.py).

Counterpart of the reference's benchmark/compare_*.py with the same knobs and the same timing recipe
(reference benchmark/compare_prefill.py:13-28: warm-up calls, then per-iteration event pairs, mean over the
iterations), plus what its tables lack: work-normalised rates (TFLOP/s, GB/s) and the fraction of the MI355X
roofline, and a JSON dump next to the optional chart.

`flash_attn` resolves to the real wheel if one is installed, otherwise to the torch-math comparator under
testsupport/ (fast mode: torch's fused scaled_dot_product_attention where the mask allows it) -- the column is
labelled with whichever was found.
"""
import json
import os
import sys

import torch

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_ROOT, "mini-flash-attention_amd"))

MFMA_PEAK_TFLOPS = 2500.0  # dense fp16/bf16, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBPS = 8000.0


def comparator():
    """-> (module, label).  The official flash_attn if importable, else the shim."""
    try:
        import flash_attn  # noqa: F401
    except ImportError:
        sys.path.insert(0, os.path.join(_ROOT, "testsupport"))
        import flash_attn
    shim = "shim" in getattr(flash_attn, "__version__", "")
    if shim:
        os.environ.setdefault("FLASH_ATTN_SHIM_FAST", "1")
    return flash_attn, ("torch SDPA (flash_attn shim)" if shim else f"flash-attn {flash_attn.__version__}")


def event_timed_ms(fn, warmup, iters):
    """Mean / median / min latency in ms of `fn()`; one event pair and one synchronise per iteration."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    samples = []
    for _ in range(iters):
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        fn()
        t1.record()
        torch.cuda.synchronize()
        samples.append(t0.elapsed_time(t1))
    samples.sort()
    return {"mean": sum(samples) / len(samples), "median": samples[len(samples) // 2], "min": samples[0]}


def prefill_flops(batch, heads, sq, sk, head_dim, causal):
    """4*B*H*Sq*Sk*D, halved when causal (the FA-paper convention BASELINE.md uses)."""
    f = 4.0 * batch * heads * sq * sk * head_dim
    return f / 2 if causal else f


def decode_bytes(batch, heads, kv_heads, skv, head_dim, elt=2):
    """K + V read once per KV head, plus q and o."""
    return 2.0 * batch * skv * kv_heads * head_dim * elt + 2.0 * batch * heads * head_dim * elt


def dtype_of(name):
    return {"float16": torch.float16, "bfloat16": torch.bfloat16}[name]


def parse_int_list(raw):
    return [int(tok) for tok in raw.split(",") if tok.strip()]


def print_table(header, rows):
    widths = [max(len(str(r[i])) for r in [header] + rows) for i in range(len(header))]
    line = " | ".join(str(h).rjust(w) for h, w in zip(header, widths))
    print(line)
    print("-" * len(line))
    for r in rows:
        print(" | ".join(str(c).rjust(w) for c, w in zip(r, widths)))


def save_outputs(results, chart_path, x_label, title, series):
    """results: list of dicts (one per x); series: [(key, label)] of the ms columns to draw as grouped bars."""
    if not chart_path:
        return
    base, _ = os.path.splitext(chart_path)
    os.makedirs(os.path.dirname(os.path.abspath(chart_path)), exist_ok=True)
    with open(base + ".json", "w") as f:
        json.dump(results, f, indent=1)
    print(f"wrote {base}.json")
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
    except ImportError:
        print("matplotlib not available: chart skipped")
        return
    fig, ax = plt.subplots(figsize=(10, 5))
    n = len(series)
    for i, (key, label) in enumerate(series):
        ax.bar([x + (i - (n - 1) / 2) * 0.8 / n for x in range(len(results))], [r[key] for r in results], 0.8 / n, label=label)
    ax.set_xticks(range(len(results)))
    ax.set_xticklabels([str(r["x"]) for r in results])
    ax.set_xlabel(x_label)
    ax.set_ylabel("latency, ms (lower is better)")
    ax.set_title(title)
    ax.legend()
    fig.tight_layout()
    fig.savefig(chart_path, dpi=160)
    plt.close(fig)
    print(f"wrote {chart_path}")
