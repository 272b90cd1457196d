"""Same-box A/B of the flash-decoding workgroup -> row mapping and of the in-kernel split merge (developer aid; needs a build
with MFA_EXTRA_HIPCC_FLAGS=-DMFA_DEV_DECODE_AB, which reads MFA_FUSED_COMBINE per launch and has no size gate on the merge):
interleaved rounds in ONE process, rotating caches, median and min of per-launch HIP-event times.
  python tools/ab_decode_map.py"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mini-flash-attention_amd"))
import mini_flash_attention as mfa


def measure(fn, iters=30):
    ts = []
    for _ in range(iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def burst(fn, n=100):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n


def shape(B, H, Hk, Sk, dt, copies, splits=0, block=None):
    D = 128
    q = torch.randn(B, 1, H, D, device="cuda", dtype=dt)
    sets = [tuple(torch.randn(B, Sk, Hk, D, device="cuda", dtype=dt) for _ in range(2)) for _ in range(copies)]
    lens = torch.full((B,), Sk, device="cuda", dtype=torch.int32)
    st = {"i": 0}

    def run():
        kc, vc = sets[st["i"] % copies]
        st["i"] += 1
        mfa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, num_splits=splits)
    return run, 2.0 * (2 * B * Sk * Hk * D + 2 * B * H * D)


def ab(name, run, by, variants, rounds=4):
    res = {v: [] for v in variants}
    for v in variants:  # warm
        os.environ.update(variants[v]); burst(run, 50)
    for r in range(rounds):
        for v in variants:
            os.environ.update(variants[v])
            burst(run, 30)
            res[v].append(burst(run, 200))
    for v in variants:
        xs = sorted(res[v])
        print(f"{name:44s} {v:18s} burst us: " + " ".join(f"{x:7.2f}" for x in res[v]) + f"  | min {xs[0]:7.2f} = {by / xs[0] / 1e3:6.0f} GB/s", flush=True)


MAP = {"row r on XCD r & 7": {}}  # (round 3a compared the XCD-contiguous order against the plain one here: profiles/r03a_ab_*)
FUSE = {"fused-merge": {"MFA_FUSED_COMBINE": "1"}, "combine-launch": {"MFA_FUSED_COMBINE": "0"}}
torch.manual_seed(0)
run, by = shape(24, 24, 8, 8192, torch.bfloat16, 2)
ab("config 3 bf16 B24 24/8 Skv8192 unsplit", run, by, MAP)
for Sk in (512, 1024, 2048, 4096):
    run, by = shape(24, 24, 8, Sk, torch.bfloat16, max(2, 1500000000 // (2 * 2 * 24 * Sk * 8 * 128)), splits=1)
    ab(f"GQA bf16 B24 24/8 Skv{Sk} splits=1", run, by, MAP)
run, by = shape(24, 24, 24, 8192, torch.float16, 2, splits=1)
ab("README MHA fp16 B24 H24 Skv8192 splits=1", run, by, MAP)
run, by = shape(24, 64, 8, 8192, torch.bfloat16, 2)
ab("packed G=8 bf16 B24 64/8 Skv8192 auto", run, by, FUSE)
run, by = shape(24, 24, 8, 8192, torch.bfloat16, 2, splits=4)
ab("config 3 forced 4 splits", run, by, FUSE)
for (B, Sk) in ((4, 8192), (8, 4096), (16, 2048), (1, 8192)):
    run, by = shape(B, 24, 8, Sk, torch.bfloat16, 8)
    ab(f"GQA bf16 B{B} 24/8 Skv{Sk} auto", run, by, FUSE)
for (B, Sk) in ((24, 512), (24, 1024), (24, 2048)):
    run, by = shape(B, 24, 24, Sk, torch.float16, 8)
    ab(f"README MHA fp16 B{B} H24 Skv{Sk} auto", run, by, FUSE)
