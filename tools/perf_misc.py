"""Perf sanity over the non-headline shapes: other head dims, GQA, varlen (BASELINE config 4), paged decode (config 5)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mini-flash-attention_amd"))
import mini_flash_attention as mfa
from perf_sweep import measure
torch.manual_seed(0)
dev = "cuda"
for (B, S, H, Hk, D, dt) in ((16, 2048, 16, 16, 32, torch.float16), (16, 2048, 16, 16, 64, torch.float16), (16, 2048, 16, 4, 64, torch.bfloat16),
                            (16, 2048, 16, 16, 96, torch.float16), (8, 2048, 16, 16, 256, torch.float16), (8, 2048, 16, 2, 256, torch.bfloat16)):
    q = torch.randn(B, S, H, D, device=dev, dtype=dt)
    k, v = (torch.randn(B, S, Hk, D, device=dev, dtype=dt) for _ in range(2))
    for causal in (True, False):
        med, mn = measure(lambda: mfa.flash_attn_func(q, k, v, causal=causal), iters=10)
        fl = 4.0 * B * H * S * S * D * (0.5 if causal else 1.0)
        print(f"prefill {str(dt)[6:]} B{B} S{S} H{H}/{Hk} D{D} causal={int(causal)}: {med:7.3f} ms {fl/med/1e9:7.1f} TFLOP/s", flush=True)
# BASELINE config 4: varlen fp16 cu_seqlens=[0,128,384,896] H8 D64 causal
lens = [128, 256, 512]
cu = torch.tensor([0] + lens, device=dev).cumsum(0).int()
q, k, v = (torch.randn(sum(lens), 8, 64, device=dev, dtype=torch.float16) for _ in range(3))
med, mn = measure(lambda: mfa.flash_attn_varlen_func(q, k, v, cu, cu, 512, 512, causal=True))
print(f"varlen config 4 (896 tokens, H8 D64 causal): med {med*1e3:.1f} us  min {mn*1e3:.1f} us", flush=True)
# a serving-sized varlen batch: 64 sequences of 100..2000 tokens, H24/8 D128
g = torch.Generator().manual_seed(1)
lens = torch.randint(100, 2000, (64,), generator=g).tolist()
cu = torch.tensor([0] + lens, device=dev).cumsum(0).int()
q = torch.randn(sum(lens), 24, 128, device=dev, dtype=torch.bfloat16)
k, v = (torch.randn(sum(lens), 8, 128, device=dev, dtype=torch.bfloat16) for _ in range(2))
med, mn = measure(lambda: mfa.flash_attn_varlen_func(q, k, v, cu, cu, max(lens), max(lens), causal=True))
fl = sum(4.0 * 24 * n * n * 128 * 0.5 for n in lens)
print(f"varlen 64 seqs 100..2000 tokens H24/8 D128 bf16 causal: {med:.3f} ms {fl/med/1e9:.1f} TFLOP/s", flush=True)
# BASELINE config 5: paged decode bf16 B16 Skv4096 page256 Hq24 Hkv8 D128 (3 rotating cache sets: 268 MB each > MALL/3)
B, Sk, H, Hk, D, page = 16, 4096, 24, 8, 128, 256
nb = Sk // page
sets = []
for i in range(3):
    kp, vp = (torch.randn(B * nb, page, Hk, D, device=dev, dtype=torch.bfloat16) for _ in range(2))
    table = torch.randperm(B * nb, device=dev).int().view(B, nb)
    sets.append((kp, vp, table))
q = torch.randn(B, 1, H, D, device=dev, dtype=torch.bfloat16)
lens_t = torch.full((B,), Sk, device=dev, dtype=torch.int32)
by = 2.0 * (2 * B * Sk * Hk * D + 2 * B * H * D)
state = {"i": 0}
def run():
    kp, vp, table = sets[state["i"] % 3]; state["i"] += 1
    mfa.flash_attn_with_kvcache(q, kp, vp, cache_seqlens=lens_t, block_table=table)
med, mn = measure(run, iters=30)
print(f"paged decode config 5 (B16 Skv4096 page256 24/8 D128, rotating 3 caches): med {med*1e3:.1f} us {by/med/1e6:.0f} GB/s", flush=True)
kc, vc = (torch.randn(B, Sk, Hk, D, device=dev, dtype=torch.bfloat16) for _ in range(2))
med, mn = measure(lambda: mfa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens_t), iters=30)
print(f"dense decode same shape (one cache, 268 MB ~ Infinity Cache): med {med*1e3:.1f} us {by/med/1e6:.0f} GB/s (effective)", flush=True)
for (Hq, Hkk) in ((8, 1), (32, 8), (64, 8)):
    qq = torch.randn(24, 1, Hq, 128, device=dev, dtype=torch.bfloat16)
    kk, vv = (torch.randn(24, 8192, Hkk, 128, device=dev, dtype=torch.bfloat16) for _ in range(2))
    ll = torch.full((24,), 8192, device=dev, dtype=torch.int32)
    byy = 2.0 * (2 * 24 * 8192 * Hkk * 128 + 2 * 24 * Hq * 128)
    med, mn = measure(lambda: mfa.flash_attn_with_kvcache(qq, kk, vv, cache_seqlens=ll), iters=20)
    print(f"decode bf16 B24 Skv8192 {Hq}/{Hkk} D128: med {med*1e3:.1f} us {byy/med/1e6:.0f} GB/s", flush=True)

# ---- §8(f) rows: sliding window, kv-cache append, multi-query kv-cache attention -----------------------------
B, S, H, D = 16, 8192, 24, 128
q, k, v = (torch.randn(B, S, H, D, device=dev, dtype=torch.float16) for _ in range(3))
for W in (256, 1024, 4096):
    med, mn = measure(lambda: mfa.flash_attn_func(q, k, v, causal=True, window_size=(W, 0)), iters=10)
    pairs = sum(min(r, W) + 1 for r in range(S))
    fl = 4.0 * B * H * pairs * D
    print(f"sliding window causal fp16 B{B} S{S} H{H} D{D} left={W}: {med:7.3f} ms  {fl/med/1e9:6.1f} TFLOP/s (visible pairs only)", flush=True)
med, mn = measure(lambda: mfa.flash_attn_func(q, k, v, causal=True), iters=5)
print(f"   full causal same shape: {med:7.3f} ms  {4.0*B*H*S*S*D*0.5/med/1e9:6.1f} TFLOP/s", flush=True)
del q, k, v
B, Sk, Hq, Hk, D, Sn = 24, 8192, 24, 8, 128, 8
kc, vc = (torch.randn(B, Sk, Hk, D, device=dev, dtype=torch.bfloat16) for _ in range(2))
lens = torch.full((B,), Sk - 64, device=dev, dtype=torch.int32)
for Sn in (1, 8, 64):
    qn = torch.randn(B, Sn, Hq, D, device=dev, dtype=torch.bfloat16)
    kn, vn = (torch.randn(B, Sn, Hk, D, device=dev, dtype=torch.bfloat16) for _ in range(2))
    med, mn = measure(lambda: mfa.flash_attn_with_kvcache(qn, kc, vc, cache_seqlens=lens, causal=True, k=kn, v=vn), iters=20)
    by = 2.0 * (2 * B * (Sk - 64 + Sn) * Hk * D)
    print(f"kv-cache append+attend bf16 B{B} Skv{Sk-64}+{Sn} {Hq}/{Hk} D{D} Sq={Sn}: {med*1e3:7.1f} us  {by/med/1e6:6.0f} GB/s of K+V", flush=True)
