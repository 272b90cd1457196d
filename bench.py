"""bench.py — headline measurement of the attention-forward hot path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of synthetic input, already resident in HBM:
  * headline (the JSON's metric/value): BASELINE config 2 — prefill, fp16, B=48 S=1024 H=24 D=128, causal,
    through mini_flash_attention.flash_attn_func; value = whole-job TFLOP/s (FlashAttention convention: causal
    FLOPs = 4*B*H*S^2*D / 2);
  * "decode" object: BASELINE config 3 — bf16 B=24 Sq=1 Skv=8192 Hq=24 Hkv=8 D=128, num_splits auto, timed the
    same way right after, reported as HBM GB/s of the algorithmic bytes (K+V once per KV head, + Q + O).
Beside the contract's timed regions (outside them, and BEFORE them): "sweep" — the reference benchmark's prefill shape family
(fp16 B=48 H=24 D=128, S = 256 .. 4096, causal and not), and "decode_sweep" — the reference README's MHA fp16
decode shapes (B=24 H=24 Skv = 512 .. 8192) and BASELINE config 5 (paged), each over ROTATING cache copies so no
launch finds its cache in the 256 MB Infinity Cache; "prefill_modes" — dense / packed varlen / paged K/V through the same
head-dim-128 kernel at one shape.  Every sweep entry carries two regimes: "cold" = the first 25
launches after an idle gap (clocks and power have not settled: a 25-launch run from idle measures the ramp, not the
kernel) and "steady" = after about half a second of back-to-back launches.  Because the sweeps run first, the headline /
decode / kvcache_packed regions start at settled clocks whatever --steps / --warmup are.
Multi-GPU: the path is embarrassingly parallel over batch x heads and has no exchange step ("replicas only",
DESIGN.md): every rank runs its own batch (weak scaling), the timed region is bracketed by barrier +
synchronize, elapsed = MAX over ranks, value = all ranks' work / that time.
`roofline` is for the dominant kernel of the headline step; `cpu_baseline` (rank 0, N=1 only) times the
reference tests' own oracle — eager torch SDPA in fp32 on the host cores — on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "mini-flash-attention_amd"))

PEAK_MFMA_TFLOPS = 2500.0  # dense fp16/bf16, MI355X_MICROARCH.md chip table
PEAK_HBM_GBPS = 8000.0     # HBM3E spec (≈6300 achievable copy)

PREFILL = dict(B=48, S=1024, H=24, Hk=24, D=128, causal=True, dtype=torch.float16)
DECODE = dict(B=24, Sk=8192, H=24, Hk=8, D=128, dtype=torch.bfloat16)
SWEEP_S = (256, 512, 1024, 2048, 4096)
DECODE_SWEEP_SK = (512, 1024, 2048, 4096, 8192)
PAGED = dict(B=16, Sk=4096, H=24, Hk=8, D=128, page=256, dtype=torch.bfloat16)


def prefill_flops(c):
    return 4.0 * c["B"] * c["H"] * c["S"] * c["S"] * c["D"] * (0.5 if c["causal"] else 1.0)


def prefill_bytes(c):
    return 2.0 * (2 * c["B"] * c["S"] * c["H"] * c["D"] + 2 * c["B"] * c["S"] * c["Hk"] * c["D"])


def decode_bytes(c):
    return 2.0 * (2 * c["B"] * c["Sk"] * c["Hk"] * c["D"] + 2 * c["B"] * c["H"] * c["D"])


def measured_traffic(key):
    """Per-launch HBM bytes from the last committed rocprofv3 PMC passes (profiles/traffic.json), or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            return json.load(f).get(key)
    except (OSError, ValueError):
        return None


def timed_region(fn, steps, warmup, dist):
    """W untimed steps, then exactly K steps between barrier+synchronize pairs.  Returns (wall seconds MAX over
    ranks, mean device ms per step from HIP events on the launch stream)."""
    for _ in range(warmup):
        fn()
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if dist:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    start.record()
    for _ in range(steps):
        fn()
    end.record()
    torch.cuda.synchronize()
    if dist:
        torch.distributed.barrier()
    wall = time.perf_counter() - t0
    ev_ms = start.elapsed_time(end) / steps
    if dist:
        t = torch.tensor([wall, ev_ms], device="cuda", dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        wall, ev_ms = t[0].item(), t[1].item()
    return wall, ev_ms


def event_ms(fn, n):
    """mean device ms per launch over n back-to-back launches (HIP events on the launch stream)"""
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(n):
        fn()
    end.record()
    torch.cuda.synchronize()
    return start.elapsed_time(end) / n


def cold_and_steady(fn, cold_n=25, settle_s=0.5, steady_n=40, idle_s=0.25):
    """(cold ms, steady ms): the first cold_n launches after an idle gap, then steady_n launches after settle_s of
    back-to-back launches.  One launch beforehand loads the code object and is not timed."""
    fn()
    torch.cuda.synchronize()
    time.sleep(idle_s)
    cold = event_ms(fn, cold_n)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < settle_s:
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
    return cold, event_ms(fn, steady_n)


def cpu_baseline(c, budget_s=20.0):
    """Eager torch SDPA, fp32, on the host cores (what reference tests/test_mha.py:75-81 uses as its oracle),
    on a batch slice of the headline workload sized to ~budget_s of CPU work."""
    threads = torch.get_num_threads()
    g = torch.Generator().manual_seed(0)
    b = 1
    q, k, v = (torch.randn(b, c["H"], c["S"], c["D"], generator=g) for _ in range(3))
    t = time.perf_counter()
    F.scaled_dot_product_attention(q, k, v, is_causal=c["causal"])
    per_b = time.perf_counter() - t
    b = int(max(1, min(c["B"], budget_s / max(per_b, 1e-4) / 4)))
    q, k, v = (torch.randn(b, c["H"], c["S"], c["D"], generator=g) for _ in range(3))
    F.scaled_dot_product_attention(q, k, v, is_causal=c["causal"])
    times = []
    for _ in range(3):
        t = time.perf_counter()
        F.scaled_dot_product_attention(q, k, v, is_causal=c["causal"])
        times.append(time.perf_counter() - t)
    med = sorted(times)[1]
    flops = prefill_flops(dict(c, B=b))
    return {
        "value": round(flops / med / 1e12, 4), "unit": "TFLOP/s", "cores": threads, "kind": "port",
        "sample": f"torch SDPA eager fp32 on host CPU, B={b} slice of the headline workload (H={c['H']} S={c['S']} "
                  f"D={c['D']} causal), median of 3 runs, {med * 1e3:.1f} ms/run, os.cpu_count()={os.cpu_count()}",
    }


def prefill_sweep(mfa, dev):
    """fp16 B=48 H=24 D=128, S in SWEEP_S, causal and not: TFLOP/s and both roofline fractions, cold and steady; then the same
    kernel in bf16 at the headline shape (S=1024 causal) and at its MFMA-bound end (S=4096 non-causal)"""
    out = []
    shapes = [(S, torch.float16, (True, False)) for S in SWEEP_S] + [(1024, torch.bfloat16, (True,)), (4096, torch.bfloat16, (False,))]
    for S, dtype, causals in shapes:
        c = dict(PREFILL, S=S, dtype=dtype)
        q, k, v = (torch.randn(c["B"], S, c["H"], c["D"], device=dev, dtype=torch.float32).to(c["dtype"]) for _ in range(3))
        for causal in causals:
            c["causal"] = causal
            n = max(4, min(40, int(0.25 / (prefill_flops(c) / 0.9e15))))
            cold, steady = cold_and_steady(lambda: mfa.flash_attn_func(q, k, v, causal=causal), steady_n=n,
                                           settle_s=max(0.5, 25 * prefill_flops(c) / 0.9e15))
            ent = {"S": S, "causal": causal, "dtype": "bf16" if dtype == torch.bfloat16 else "f16"}
            for nm, ms in (("cold", cold), ("steady", steady)):
                tf, gb = prefill_flops(c) / ms / 1e9, prefill_bytes(c) / ms / 1e6
                ent[nm] = {"ms": round(ms, 4), "tflops": round(tf, 1), "mfma_frac": round(tf / PEAK_MFMA_TFLOPS, 4),
                           "hbm_gbps": round(gb, 1), "hbm_frac": round(gb / PEAK_HBM_GBPS, 4)}
            out.append(ent)
        del q, k, v
    return out


def prefill_modes(mfa, dev):
    """The non-dense ways into the head-dim-128 prefill kernel at one shape (bf16, 16 sequences x 2048, H 24/8, causal): dense
    (B,S,H,D), packed variable-length (cu_seqlens) and the same over a paged K/V cache (page 256, permuted block table); steady
    TFLOP/s and the ratio to the dense launch"""
    B, S, H, Hk, D, page = 16, 2048, 24, 8, 128, 256
    flops = 4.0 * B * H * S * S * D * 0.5
    q = torch.randn(B * S, H, D, device=dev, dtype=torch.float32).to(torch.bfloat16)
    kd, vd = (torch.randn(B, S, Hk, D, device=dev, dtype=torch.float32).to(torch.bfloat16) for _ in range(2))
    cu = torch.arange(0, (B + 1) * S, S, device=dev, dtype=torch.int32)
    nb = B * S // page
    perm = torch.randperm(nb, generator=torch.Generator().manual_seed(0)).to(dev)
    kp, vp = torch.empty(nb, page, Hk, D, device=dev, dtype=torch.bfloat16), torch.empty(nb, page, Hk, D, device=dev, dtype=torch.bfloat16)
    kp[perm], vp[perm] = kd.view(nb, page, Hk, D), vd.view(nb, page, Hk, D)
    table = perm.int().view(B, S // page).contiguous()
    runs = (("dense", lambda: mfa.flash_attn_func(q.view(B, S, H, D), kd, vd, causal=True)),
            ("varlen", lambda: mfa.flash_attn_varlen_func(q, kd.view(B * S, Hk, D), vd.view(B * S, Hk, D), cu, cu, S, S, causal=True)),
            ("paged_page256", lambda: mfa.flash_attn_varlen_func(q, kp, vp, cu, cu, S, S, causal=True, block_table=table)))
    out = {"workload": "bf16 16 x 2048 tokens, Hq24 Hkv8 D128, causal"}
    for name, fn in runs:
        _, steady = cold_and_steady(fn, steady_n=20, settle_s=0.5)
        out[name] = {"ms": round(steady, 4), "tflops": round(flops / steady / 1e9, 1)}
    for name in ("varlen", "paged_page256"):
        out[name]["vs_dense"] = round(out["dense"]["ms"] / out[name]["ms"], 3)
    # a ragged batch: 16 sequences of 512 .. 4096 tokens (seeded), packed varlen, causal
    lens = torch.randint(512, 4097, (16,), generator=torch.Generator().manual_seed(1)).tolist()
    tot = sum(lens)
    qr = torch.randn(tot, H, D, device=dev, dtype=torch.float32).to(torch.bfloat16)
    kr, vr = (torch.randn(tot, Hk, D, device=dev, dtype=torch.float32).to(torch.bfloat16) for _ in range(2))
    cur = torch.tensor([0] + lens, device=dev).cumsum(0).int()
    _, steady = cold_and_steady(lambda: mfa.flash_attn_varlen_func(qr, kr, vr, cur, cur, max(lens), max(lens), causal=True), steady_n=20, settle_s=0.5)
    out["varlen_ragged"] = {"workload": f"bf16 16 sequences of 512 .. 4096 tokens ({tot} in all), Hq24 Hkv8 D128, causal", "ms": round(steady, 4),
                            "tflops": round(sum(4.0 * H * n * n * D * 0.5 for n in lens) / steady / 1e9, 1)}
    return out


def decode_sweep(mfa, dev):
    """README MHA fp16 decode shapes (B=24 H=24 D=128 Sq=1) and BASELINE config 5 (paged), over rotating cache copies"""
    out = []
    for Sk in DECODE_SWEEP_SK:
        c = dict(B=24, Sk=Sk, H=24, Hk=24, D=128)
        copies = max(2, min(8, -(-int(1.5e9) // int(decode_bytes(c)))))
        sets = [tuple(torch.randn(c["B"], Sk, c["Hk"], c["D"], device=dev, dtype=torch.float16) for _ in range(2)) for _ in range(copies)]
        q = torch.randn(c["B"], 1, c["H"], c["D"], device=dev, dtype=torch.float16)
        lens = torch.full((c["B"],), Sk, dtype=torch.int32, device=dev)
        st = {"i": 0}

        def run():
            kc, vc = sets[st["i"] % copies]
            st["i"] += 1
            mfa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, num_splits=0)

        cold, steady = cold_and_steady(run, steady_n=40, settle_s=0.3)
        ent = {"workload": f"decode fp16 MHA B=24 Sq=1 Skv={Sk} H=24 D=128 num_splits=auto", "rotating_copies": copies}
        for nm, ms in (("cold", cold), ("steady", steady)):
            gb = decode_bytes(c) / ms / 1e6
            ent[nm] = {"us": round(ms * 1e3, 2), "hbm_gbps": round(gb, 1), "hbm_frac": round(gb / PEAK_HBM_GBPS, 4)}
        out.append(ent)
        del sets, q
    p = PAGED
    nb = p["Sk"] // p["page"]
    sets = []
    for _ in range(4):
        kp, vp = (torch.randn(p["B"] * nb, p["page"], p["Hk"], p["D"], device=dev, dtype=p["dtype"]) for _ in range(2))
        sets.append((kp, vp, torch.randperm(p["B"] * nb, device=dev).int().view(p["B"], nb)))
    q = torch.randn(p["B"], 1, p["H"], p["D"], device=dev, dtype=p["dtype"])
    lens = torch.full((p["B"],), p["Sk"], dtype=torch.int32, device=dev)
    st = {"i": 0}

    def run_paged():
        kp, vp, table = sets[st["i"] % 4]
        st["i"] += 1
        mfa.flash_attn_with_kvcache(q, kp, vp, cache_seqlens=lens, block_table=table)

    cold, steady = cold_and_steady(run_paged, steady_n=40, settle_s=0.3)
    ent = {"workload": "paged decode bf16 B=16 Sq=1 Skv=4096 Hq=24 Hkv=8 D=128 page=256, permuted block table (BASELINE config 5)",
           "rotating_copies": 4}
    for nm, ms in (("cold", cold), ("steady", steady)):
        gb = decode_bytes(p) / ms / 1e6
        ent[nm] = {"us": round(ms * 1e3, 2), "hbm_gbps": round(gb, 1), "hbm_frac": round(gb / PEAK_HBM_GBPS, 4)}
    out.append(ent)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: long enough for the steady state (the first ~100 launches after an idle period run 5-9 % slower on
    # MI355X while clocks and power settle; 350 launches of 0.5 ms are still well under a second)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sweep", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # (MFA_BENCH_FORCE_DIST=1: take the RCCL path with a single rank, to rehearse it on a one-GPU box)
    dist = world > 1 or os.environ.get("MFA_BENCH_FORCE_DIST") == "1"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if dist:
        torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    n_gpus = world if dist else 1

    import mini_flash_attention as mfa  # fails loudly when the HIP extension is missing

    dev = torch.device("cuda", local_rank)
    torch.manual_seed(rank)
    # The sweeps run FIRST (every rank runs its own): they carry both regimes per shape, and they leave the GPU at settled
    # clocks, so the contract's timed regions below (W warm-up + K timed steps each) measure the kernels rather than the
    # power-state ramp of the first ~30 ms after an idle period -- which `sweep[*].cold` reports, shape by shape.
    sweeps = {}
    if not args.no_sweep:
        sweeps["decode_sweep"] = decode_sweep(mfa, dev)
        sweeps["prefill_modes"] = prefill_modes(mfa, dev)
        sweeps["sweep"] = prefill_sweep(mfa, dev)  # (last: the headline region follows the MFMA-heavy shapes directly)
        sweeps["regimes"] = ("cold = mean of the first 25 launches after a 0.25 s idle gap; steady = mean of 4-40 launches "
                             "after >= 0.3-0.5 s of back-to-back launches; HIP events on the launch stream; the headline, decode "
                             "and kvcache_packed regions run after the sweeps (settled clocks)")
    c = PREFILL
    q = torch.randn(c["B"], c["S"], c["H"], c["D"], device=dev, dtype=torch.float32).to(c["dtype"])
    k = torch.randn(c["B"], c["S"], c["Hk"], c["D"], device=dev, dtype=torch.float32).to(c["dtype"])
    v = torch.randn(c["B"], c["S"], c["Hk"], c["D"], device=dev, dtype=torch.float32).to(c["dtype"])
    wall, ev_ms = timed_region(lambda: mfa.flash_attn_func(q, k, v, causal=c["causal"]), args.steps, args.warmup, dist)
    ms_per_step = wall / args.steps * 1e3
    tflops = prefill_flops(c) * n_gpus * args.steps / wall / 1e12
    kern_s = ev_ms * 1e-3
    kern_tflops = prefill_flops(c) / kern_s / 1e12
    kern_gbps = prefill_bytes(c) / kern_s / 1e9
    # binding roof = max(FLOPs / P_mfma, bytes / BW_hbm)   (config 2 sits on the ridge, HBM side)
    t_mfma = prefill_flops(c) / (PEAK_MFMA_TFLOPS * 1e12)
    t_hbm = prefill_bytes(c) / (PEAK_HBM_GBPS * 1e9)
    if t_hbm >= t_mfma:
        roof = {"bound": "hbm", "achieved": round(kern_gbps, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                "frac": round(kern_gbps / PEAK_HBM_GBPS, 4)}
    else:
        roof = {"bound": "mfma", "achieved": round(kern_tflops, 1), "peak": PEAK_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": round(kern_tflops / PEAK_MFMA_TFLOPS, 4)}
    roof.update({"traffic": measured_traffic("prefill"), "traffic_source": "profiles/traffic.json (replayed)",
                 "kernel": "prefill64_kernel<Half, dense>", "kernel_ms": round(ev_ms, 4),
                 "mfma_tflops": round(kern_tflops, 1), "mfma_frac": round(kern_tflops / PEAK_MFMA_TFLOPS, 4),
                 "hbm_gbps": round(kern_gbps, 1), "hbm_frac": round(kern_gbps / PEAK_HBM_GBPS, 4),
                 "algorithmic_flops": prefill_flops(c), "algorithmic_bytes": prefill_bytes(c)})
    del q, k, v

    d = DECODE
    qd = torch.randn(d["B"], 1, d["H"], d["D"], device=dev, dtype=torch.float32).to(d["dtype"])
    kc = torch.randn(d["B"], d["Sk"], d["Hk"], d["D"], device=dev, dtype=torch.float32).to(d["dtype"])
    vc = torch.randn(d["B"], d["Sk"], d["Hk"], d["D"], device=dev, dtype=torch.float32).to(d["dtype"])
    lens = torch.full((d["B"],), d["Sk"], dtype=torch.int32, device=dev)
    dwall, dev_ms = timed_region(lambda: mfa.flash_attn_with_kvcache(qd, kc, vc, cache_seqlens=lens, num_splits=0),
                                 args.steps, args.warmup, dist)
    dec_gbps = decode_bytes(d) * n_gpus * args.steps / dwall / 1e9
    dec_kern_gbps = decode_bytes(d) / (dev_ms * 1e-3) / 1e9
    decode = {
        "metric": "decode HBM GB/s", "value": round(dec_gbps, 1), "unit": "GB/s", "us_per_step": round(dwall / args.steps * 1e6, 2),
        "dtype": "bf16",
        "config": {"workload": "flash-decoding bf16 B=24 Sq=1 Skv=8192 Hq=24 Hkv=8 D=128 num_splits=auto (BASELINE config 3)"},
        "roofline": {"bound": "hbm", "achieved": round(dec_kern_gbps, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                     "frac": round(dec_kern_gbps / PEAK_HBM_GBPS, 4), "traffic": measured_traffic("decode"),
                     "traffic_source": "profiles/traffic.json (replayed)",
                     "kernel": "decode_split_kv_kernel<BFloat,16,3,dense> (192 workgroups, unsplit: no merge)", "kernel_us": round(dev_ms * 1e3, 2),
                     "algorithmic_bytes": decode_bytes(d)},
    }

    # beside the two BASELINE shapes: the packed-row kv-cache kernel on a GQA group of 8 (same cache, 64 query heads)
    del qd
    qg = torch.randn(d["B"], 1, 8 * d["Hk"], d["D"], device=dev, dtype=torch.float32).to(d["dtype"])
    g8 = dict(d, H=8 * d["Hk"])
    gwall, gev_ms = timed_region(lambda: mfa.flash_attn_with_kvcache(qg, kc, vc, cache_seqlens=lens, num_splits=0),
                                 args.steps, args.warmup, dist)
    packed = {
        "metric": "decode HBM GB/s, GQA group 8 (packed-row MFMA kernel)", "value": round(decode_bytes(g8) * n_gpus * args.steps / gwall / 1e9, 1),
        "unit": "GB/s", "us_per_step": round(gwall / args.steps * 1e6, 2), "dtype": "bf16",
        "config": {"workload": "kv-cache attention bf16 B=24 Sq=1 Skv=8192 Hq=64 Hkv=8 D=128 num_splits=auto"},
        "roofline": {"bound": "hbm", "achieved": round(decode_bytes(g8) / (gev_ms * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                     "frac": round(decode_bytes(g8) / (gev_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4),
                     "traffic": measured_traffic("kvcache_packed"), "traffic_source": "profiles/traffic.json (replayed)",
                     "kernel": "prefill_fwd_kernel<BFloat,128,4,0,MQ,STREAM> (key splits) + decode_combine_kernel (1 536 workgroups: "
                               "the merge stays its own launch at this size)", "kernel_us": round(gev_ms * 1e3, 2),
                     "algorithmic_bytes": decode_bytes(g8)},
    }
    del qg, kc, vc

    if rank == 0:
        out = {
            "metric": "prefill attention TFLOPS (fp16, head_dim=128, causal) + decode HBM GB/s, 1xMI355X",
            "value": round(tflops, 2), "unit": "TFLOP/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16", "data": "synthetic",
            "config": {"workload": "prefill fp16 B=48 S=1024 H=24 D=128 causal per GPU (BASELINE config 2, benchmark/prefill.py shape)",
                       "parallelism": f"replicas x{n_gpus} (no collective on the data path)"},
            "roofline": roof, "decode": decode, "kvcache_packed": packed,
        }
        # the same shape's two regimes, from the sweep (the timed region above runs at settled clocks: see the module docstring)
        for e in sweeps.get("sweep", []):
            if e["S"] == c["S"] and e["causal"] == c["causal"] and e["dtype"] == "f16":
                out["headline_regimes"] = {"cold_tflops": e["cold"]["tflops"], "steady_tflops": e["steady"]["tflops"],
                                           "note": "cold = first 25 launches after an idle gap; value = the contract's timed region, which "
                                                   "follows the sweeps and so starts at settled clocks whatever --warmup is"}
        out.update(sweeps)
        if n_gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(c)
        print(json.dumps(out), flush=True)
    if dist:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
