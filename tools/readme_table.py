"""Rewrites the bench-derived rows of README.md's results table (and the record paragraph above it) from the committed records:
profiles/r03z_bench_steps20_warmup5.json, r03z_bench_default.json, r03z_final_summary.json.   python tools/readme_table.py"""
import json, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.load(open(os.path.join(ROOT, "profiles", "r03z_bench_steps20_warmup5.json")))
dd = json.load(open(os.path.join(ROOT, "profiles", "r03z_bench_default.json")))
sm = json.load(open(os.path.join(ROOT, "profiles", "r03z_final_summary.json")))["kernels"]
sw = {(e["S"], e["causal"], e["dtype"]): e for e in d["sweep"]}
ds = d["decode_sweep"]
pm = d["prefill_modes"]
tf = lambda e, r: round(e[r]["tflops"])
pk = next(v for k, v in sm.items() if "prefill64_kernel" in k)
dk = next(v for k, v in sm.items() if "decode_split_kv" in k)
rows = {
    "| prefill, headline |": f"| prefill, headline | fp16 B48 S1024 H24 D128 causal | {round(d['headline_regimes']['cold_tflops'])} TFLOP/s | {round(d['headline_regimes']['steady_tflops'])} TFLOP/s ({round(d['value'])} in the 20-step timed region, {d['ms_per_step']:.3f} ms; {round(dd['value'])} over 300 steps) | {100 * d['roofline']['mfma_frac']:.0f} % of 2.5 PF dense MFMA, {100 * sw[(1024, True, 'f16')]['steady']['hbm_frac']:.0f} % of 8 TB/s (ridge point) |",
    "| prefill, bf16 |": f"| prefill, bf16 | same shape / S=4096 non-causal | {tf(sw[(1024, True, 'bf16')], 'cold')} / {tf(sw[(4096, False, 'bf16')], 'cold')} | {tf(sw[(1024, True, 'bf16')], 'steady')} / {tf(sw[(4096, False, 'bf16')], 'steady')} TFLOP/s | {100 * sw[(4096, False, 'bf16')]['steady']['mfma_frac']:.0f} % of MFMA peak at S=4096 |",
    "| prefill | fp16, S=2048 / 4096 non-causal |": f"| prefill | fp16, S=2048 / 4096 non-causal | {tf(sw[(2048, False, 'f16')], 'cold')} / {tf(sw[(4096, False, 'f16')], 'cold')} | {tf(sw[(2048, False, 'f16')], 'steady')} / {tf(sw[(4096, False, 'f16')], 'steady')} TFLOP/s | {100 * sw[(2048, False, 'f16')]['steady']['mfma_frac']:.0f} % / {100 * sw[(4096, False, 'f16')]['steady']['mfma_frac']:.0f} % of MFMA peak (a loop of bare MFMAs sustains 68 % on this chip) |",
    "| prefill | fp16, S=256 / 512 causal |": f"| prefill | fp16, S=256 / 512 causal | {tf(sw[(256, True, 'f16')], 'cold')} / {tf(sw[(512, True, 'f16')], 'cold')} | {tf(sw[(256, True, 'f16')], 'steady')} / {tf(sw[(512, True, 'f16')], 'steady')} TFLOP/s | {100 * sw[(256, True, 'f16')]['steady']['hbm_frac']:.0f} % / {100 * sw[(512, True, 'f16')]['steady']['hbm_frac']:.0f} % of 8 TB/s (HBM-bound sizes) |",
    "| prefill, varlen / paged |": f"| prefill, varlen / paged | bf16 16 × 2048 H24/8 causal: dense / packed varlen / paged, page 256 (`prefill_modes`) | | {pm['dense']['ms']:.3f} / {pm['varlen']['ms']:.3f} / {pm['paged_page256']['ms']:.3f} ms, {round(pm['dense']['tflops'])} / {round(pm['varlen']['tflops'])} / {round(pm['paged_page256']['tflops'])} TFLOP/s | varlen {pm['varlen']['vs_dense']:.2f} ×, paged {pm['paged_page256']['vs_dense']:.2f} × dense in this run (0.96–0.98 × over pages 64 … 1024 in one process; round 2's general kernel: 0.83 ×) |",
    "| prefill, ragged varlen |": f"| prefill, ragged varlen | bf16 H24/8 causal, 16 sequences of 512 … 4096 tokens (`prefill_modes.varlen_ragged`) | | {pm['varlen_ragged']['ms']:.3f} ms, {round(pm['varlen_ragged']['tflops'])} TFLOP/s | schedule built from the lengths; same box A/B against round 2's routing in `profiles/r03_varlen_p64_vs_general.txt`: 0.54 / 0.56 / 0.37 ms where it took 0.76 / 1.22 / 2.53 |",
    "| decode | bf16 B24 Skv8192":  f"| decode | bf16 B24 Skv8192 Hq24 Hkv8 D128 (config 3) | | {d['decode']['us_per_step']:.1f} µs, {round(d['decode']['value'])} GB/s ({dd['decode']['us_per_step']:.1f} µs, {round(dd['decode']['value'])} GB/s over 300 steps; boxes of the pool: 128–137 µs) | {100 * d['decode']['value'] / 8000:.0f}–{100 * dd['decode']['value'] / 8000:.0f} % of 8 TB/s (non-temporal streaming ceiling measured at 6.4–6.9 TB/s) |",
    "| decode, MHA |": "| decode, MHA | fp16 B24 H24 Skv 512 … 8192 (reference README shapes, rotating caches) | " + " / ".join(f"{e['cold']['hbm_gbps'] / 1e3:.2f}" for e in ds[:5]) + " TB/s | " + " / ".join(f"{e['steady']['hbm_gbps'] / 1e3:.2f}" for e in ds[:5]) + f" TB/s | {100 * ds[0]['steady']['hbm_frac']:.0f}–{100 * max(e['steady']['hbm_frac'] for e in ds[:5]):.0f} % |",
    "| decode, G=8 |": f"| decode, G=8 | bf16 B24 Skv8192 Hq64 Hkv8 D128 | | {round(d['kvcache_packed']['value'])} GB/s (packed-row MFMA kernel) | {100 * d['kvcache_packed']['value'] / 8000:.0f} % |",
    "| paged decode |": f"| paged decode | bf16 B16 Skv4096 Hq24 Hkv8 page 256 (config 5, rotating caches) | {ds[5]['cold']['us']:.1f} µs | {ds[5]['steady']['us']:.1f} µs, {round(ds[5]['steady']['hbm_gbps'])} GB/s | {100 * ds[5]['steady']['hbm_frac']:.0f} % |",
    "| CPU baseline |": f"| CPU baseline | eager torch SDPA fp32 on the box's host threads | {d['cpu_baseline']['value']:.3f} TFLOP/s ({d['cpu_baseline']['cores']} threads) | | — |",
}
path = os.path.join(ROOT, "README.md")
lines = open(path).read().split("\n")
for i, ln in enumerate(lines):
    for key, new in rows.items():
        if ln.startswith(key):
            lines[i] = new
txt = "\n".join(lines)
ghz = pk["shader_cycles"] / pk["avg_us"] / 1e3
sp = lambda n: f"{round(n):,}".replace(",", " ")
para = (f"TFLOP/s and 5 880 – 6 280 GB/s — the last one, {round(d['value'])} TFLOP/s and {sp(d['decode']['value'])} GB/s, is `profiles/r03z_bench_steps20_warmup5.json`; default 300\n"
        f"steps: `profiles/r03z_bench_default.json`, {round(dd['value'])} TFLOP/s and {sp(dd['decode']['value'])} GB/s; rocprofv3 summary `profiles/r03z_final_*`: prefill kernel {pk['avg_us']:.1f} µs\n"
        f"average over {pk['calls']} launches at {ghz:.2f} GHz, HBM traffic {pk['hbm_traffic_bytes'] / 1e9:.2f} GB = {pk['hbm_traffic_bytes'] / 1207959552:.2f} × algorithmic, decode kernel {dk['avg_us']:.1f} µs): ")
txt = re.sub(r"TFLOP/s and 5 880 – 6 280 GB/s — the last one.*?decode kernel [0-9.]+ µs[^)]*\): ", lambda m: para, txt, flags=re.S)
open(path, "w").write(txt)
print("README.md rows rewritten from the records")
