"""Developer smoke check on a GPU box: every mode vs torch SDPA (fp32, on the GPU), prints max/mean error.
Not a test (tests/ holds the parity suite); used while iterating on kernels:  gpurun -- python tools/gpu_quickcheck.py
"""
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mini-flash-attention_amd"))
import mini_flash_attention as mfa  # noqa: E402


def sdpa(q, k, v, causal):
    g = q.shape[2] // k.shape[2]
    qf, kf, vf = (t.float().transpose(1, 2) for t in (q, k, v))
    kf = kf.repeat_interleave(g, dim=1)
    vf = vf.repeat_interleave(g, dim=1)
    return F.scaled_dot_product_attention(qf, kf, vf, is_causal=causal).transpose(1, 2)


def report(name, out, ref):
    d = (out.float() - ref.float()).abs()
    bad = not torch.isfinite(out.float()).all()
    print(f"{name:60s} max={d.max().item():.3e} mean={d.mean().item():.3e}{'  NONFINITE' if bad else ''}", flush=True)
    return d.max().item()


def main():
    torch.manual_seed(0)
    dev = "cuda"
    worst = 0.0
    for dt in (torch.float16, torch.bfloat16):
        for (B, Sq, Sk, H, Hk, D, causal) in [
            (2, 128, 128, 4, 4, 64, False), (1, 64, 64, 2, 2, 128, True), (2, 256, 256, 8, 2, 128, True),
            (1, 100, 100, 4, 4, 128, True), (1, 513, 513, 4, 2, 128, False), (2, 1024, 1024, 4, 4, 128, True),
            (1, 257, 257, 2, 2, 32, True), (1, 200, 200, 2, 2, 96, False), (1, 129, 129, 2, 1, 256, True),
            (1, 7, 7, 2, 2, 64, True), (1, 1, 1, 1, 1, 128, False), (2, 96, 300, 4, 2, 128, False),
        ]:
            q = torch.randn(B, Sq, H, D, device=dev, dtype=dt)
            k = torch.randn(B, Sk, Hk, D, device=dev, dtype=dt)
            v = torch.randn(B, Sk, Hk, D, device=dev, dtype=dt)
            out = mfa.flash_attn_func(q, k, v, causal=causal)
            torch.cuda.synchronize()
            worst = max(worst, report(f"prefill {dt} B{B} Sq{Sq} Sk{Sk} H{H}/{Hk} D{D} causal={causal}", out,
                                      sdpa(q, k, v, causal)))
        # varlen
        seqlens = [128, 256, 512]
        H, Hk, D = 8, 8, 64
        tot = sum(seqlens)
        q = torch.randn(tot, H, D, device=dev, dtype=dt)
        k = torch.randn(tot, Hk, D, device=dev, dtype=dt)
        v = torch.randn(tot, Hk, D, device=dev, dtype=dt)
        cu = torch.tensor([0] + seqlens, device=dev).cumsum(0).int()
        out = mfa.flash_attn_varlen_func(q, k, v, cu, cu, max(seqlens), max(seqlens), causal=True)
        ref = torch.cat([sdpa(q[a:b][None], k[a:b][None], v[a:b][None], True)[0]
                         for a, b in zip(cu[:-1].tolist(), cu[1:].tolist())])
        worst = max(worst, report(f"varlen {dt} {seqlens} H{H} D{D} causal", out, ref))
        # decode
        for (B, Sk, H, Hk, D, splits, paged, page) in [
            (2, 256, 4, 4, 128, 1, False, 0), (2, 1000, 8, 2, 128, 0, False, 0), (3, 8192, 24, 8, 128, 0, False, 0),
            (2, 2048, 8, 2, 64, 2, False, 0), (2, 777, 8, 1, 128, 3, False, 0), (2, 1024, 16, 1, 128, 0, False, 0),
            (2, 257, 4, 4, 256, 1, False, 0), (4, 4096, 24, 8, 128, 0, True, 256), (2, 300, 8, 2, 128, 2, True, 16),
            (2, 65, 2, 2, 32, 1, False, 0), (2, 640, 6, 2, 96, 0, False, 0),
        ]:
            q = torch.randn(B, 1, H, D, device=dev, dtype=dt)
            kc = torch.randn(B, Sk, Hk, D, device=dev, dtype=dt)
            vc = torch.randn(B, Sk, Hk, D, device=dev, dtype=dt)
            lens = torch.randint(max(1, Sk // 2), Sk + 1, (B,), device=dev, dtype=torch.int32)
            lens[0] = Sk
            if paged:
                nb = (Sk + page - 1) // page
                perm = torch.randperm(B * nb, device=dev).int().view(B, nb)
                kp = torch.zeros(B * nb, page, Hk, D, device=dev, dtype=dt)
                vp = torch.zeros_like(kp)
                pad = nb * page - Sk
                kpad = F.pad(kc, (0, 0, 0, 0, 0, pad)).view(B, nb, page, Hk, D)
                vpad = F.pad(vc, (0, 0, 0, 0, 0, pad)).view(B, nb, page, Hk, D)
                kp[perm.long().view(-1)] = kpad.view(B * nb, page, Hk, D)
                vp[perm.long().view(-1)] = vpad.view(B * nb, page, Hk, D)
                out = mfa.flash_attn_with_kvcache(q, kp, vp, cache_seqlens=lens, block_table=perm, num_splits=splits)
            else:
                out = mfa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, num_splits=splits)
            ref = torch.cat([sdpa(q[i:i + 1], kc[i:i + 1, :lens[i]], vc[i:i + 1, :lens[i]], False) for i in range(B)])
            worst = max(worst, report(f"decode {dt} B{B} Sk{Sk} H{H}/{Hk} D{D} splits={splits} paged={paged}/{page}",
                                      out, ref))
    print("WORST", worst)

    # quick timing
    def timeit(fn, n=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n

    B, S, H, D = 48, 1024, 24, 128
    q, k, v = (torch.randn(B, S, H, D, device=dev, dtype=torch.float16) for _ in range(3))
    for causal in (True, False):
        t = timeit(lambda: mfa.flash_attn_func(q, k, v, causal=causal))
        fl = 4 * B * H * S * S * D * (0.5 if causal else 1.0)
        print(f"prefill fp16 B48 S1024 H24 D128 causal={causal}: {t*1e3:.3f} ms  {fl/t/1e12:.1f} TFLOP/s", flush=True)
        qq = q.transpose(1, 2)
        t2 = timeit(lambda: F.scaled_dot_product_attention(qq, k.transpose(1, 2), v.transpose(1, 2), is_causal=causal), 5)
        print(f"   torch SDPA (GPU) same shape: {t2*1e3:.3f} ms  {fl/t2/1e12:.1f} TFLOP/s", flush=True)
    B, Sk, H, Hk = 24, 8192, 24, 8
    q = torch.randn(B, 1, H, D, device=dev, dtype=torch.bfloat16)
    kc = torch.randn(B, Sk, Hk, D, device=dev, dtype=torch.bfloat16)
    vc = torch.randn(B, Sk, Hk, D, device=dev, dtype=torch.bfloat16)
    lens = torch.full((B,), Sk, device=dev, dtype=torch.int32)
    by = 2 * B * Sk * Hk * D * 2 + 2 * B * H * D * 2
    for splits in (0, 1, 2, 4, 8, 16, 32):
        t = timeit(lambda: mfa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, num_splits=splits))
        print(f"decode bf16 B24 Skv8192 24/8 D128 splits={splits}: {t*1e6:.1f} us  {by/t/1e9:.0f} GB/s", flush=True)


if __name__ == "__main__":
    main()
