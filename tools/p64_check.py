"""Quick parity + timing check of the 64-rows-per-wave prefill kernel (dense and packed varlen) against SDPA-fp32 (GPU box).
  python tools/p64_check.py [quick]"""
import os, subprocess, sys, time
import torch
os.environ.setdefault("MFA_PREFILL64", "2")  # every shape the 64-row kernel serves goes to it
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mini-flash-attention_amd"))
import mini_flash_attention as mfa
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _knobs; _knobs.apply()
import torch.nn.functional as F
from torch.nn.attention import SDPBackend, sdpa_kernel

def ref(q, k, v, causal):
    g = q.size(2) // k.size(2)
    qf, kf, vf = (t.float().transpose(1, 2) for t in (q, k, v))
    if g > 1:
        kf, vf = kf.repeat_interleave(g, 1), vf.repeat_interleave(g, 1)
    with sdpa_kernel(SDPBackend.MATH):
        return F.scaled_dot_product_attention(qf, kf, vf, is_causal=causal).transpose(1, 2)

torch.manual_seed(0)
bad = 0
shapes = [(1, 64, 64, 1, 1), (1, 256, 256, 2, 2), (2, 512, 512, 4, 2), (1, 1024, 1024, 2, 2), (1, 200, 200, 2, 1), (1, 300, 333, 3, 3),
          (2, 1, 77, 2, 2), (1, 1000, 1000, 2, 2), (1, 2048, 2048, 1, 1), (1, 257, 511, 2, 2), (3, 129, 64, 2, 2)]
for dt in (torch.float16, torch.bfloat16):
    for (B, Sq, Sk, H, Hk) in shapes:
        for causal in (False, True):
            if causal and Sq != Sk:
                continue
            q = torch.randn(B, Sq, H, 128, device="cuda").to(dt)
            k = torch.randn(B, Sk, Hk, 128, device="cuda").to(dt)
            v = torch.randn(B, Sk, Hk, 128, device="cuda").to(dt)
            o = mfa.flash_attn_func(q, k, v, causal=causal)
            torch.cuda.synchronize()
            r = ref(q, k, v, causal)
            err = (o.float() - r).abs()
            tol = 1e-3 + (1e-3 + (2 ** -11 if dt == torch.float16 else 2 ** -8)) * r.abs() + (3e-3 if dt == torch.bfloat16 else 0)
            ok = bool((err <= tol).all()) and bool(torch.isfinite(o).all())
            bad += not ok
            print(f"{str(dt)[6:]:9s} B{B} Sq{Sq} Sk{Sk} H{H}/{Hk} causal={int(causal)}: max {err.max().item():.2e} mean {err.mean().item():.2e} {'ok' if ok else 'FAIL'}", flush=True)
# packed variable-length batches on the same kernel (its VL instances): per-sequence SDPA-fp32, O and LSE; ragged lengths,
# lengths below one row block, different query / key lengths with and without the (top-left) causal bound, sequences with no
# rows and with no keys; the route query says which kernel ran
from mini_flash_attention import capi
_lib = capi.load()
varlen_cases = [([64], [64]), ([1, 77, 300, 512, 1000], None), ([256] * 4, None), ([1024, 1024], None), ([100, 257], [333, 64]),
                ([300, 100], [200, 400]), ([513, 0, 129], [513, 70, 129]), ([40, 260, 5], [40, 0, 700])]
for dt in (torch.float16, torch.bfloat16):
    for (lq, lk) in varlen_cases:
        for causal in (False, True):
            lk_ = lk or lq
            if causal and lk and 0 in lk:  # (a causal row with no key is the general kernel's business; SDPA gives NaN)
                continue
            H, Hk = 4, 2
            q = torch.randn(sum(lq), H, 128, device="cuda").to(dt)
            k = torch.randn(sum(lk_), Hk, 128, device="cuda").to(dt)
            v = torch.randn(sum(lk_), Hk, 128, device="cuda").to(dt)
            cq = torch.tensor([0] + lq, device="cuda").cumsum(0).int()
            ck = torch.tensor([0] + lk_, device="cuda").cumsum(0).int()
            o, lse = mfa.flash_attn_varlen_func(q, k, v, cq, ck, max(lq), max(lk_), causal=causal, return_softmax_lse=True)
            route = _lib.mfa_debug_last_route()
            torch.cuda.synchronize()
            assert route & capi.MFA_ROUTE_PREFILL64, route
            ok, worst = bool(torch.isfinite(o).all()), 0.0
            for i in range(len(lq)):
                q0, k0 = int(cq[i]), int(ck[i])
                if lq[i] == 0:
                    continue
                oi, li = o[q0:q0 + lq[i]].float(), lse[:, q0:q0 + lq[i]]
                if lk_[i] == 0:
                    ok = ok and bool((oi == 0).all()) and bool(torch.isinf(li).all() and (li < 0).all())
                    continue
                qf = q[q0:q0 + lq[i]].float().transpose(0, 1)
                kf = k[k0:k0 + lk_[i]].float().transpose(0, 1).repeat_interleave(H // Hk, 0)
                vf = v[k0:k0 + lk_[i]].float().transpose(0, 1).repeat_interleave(H // Hk, 0)
                sc = qf @ kf.transpose(1, 2) / 128 ** 0.5
                if causal:  # top-left: key <= row
                    sc = sc.masked_fill(torch.arange(lk_[i], device="cuda")[None, :] > torch.arange(lq[i], device="cuda")[:, None], float("-inf"))
                r = (torch.softmax(sc, -1) @ vf).transpose(0, 1)
                err = (oi - r).abs()
                tol = 1e-3 + (1e-3 + (2 ** -11 if dt == torch.float16 else 2 ** -8)) * r.abs() + (3e-3 if dt == torch.bfloat16 else 0)
                ok = ok and bool((err <= tol).all()) and bool((li - torch.logsumexp(sc, -1)).abs().max() < 2e-3)
                worst = max(worst, err.max().item())
            # a sequence's rows do not depend on what else is in the batch or on the schedule that dealt out its row blocks:
            # each sequence once more as a batch of one, bit for bit
            for i in range(len(lq)):
                if lq[i] == 0 or lk_[i] == 0:
                    continue
                q0, k0 = int(cq[i]), int(ck[i])
                c1q, c1k = torch.tensor([0, lq[i]], device="cuda").int(), torch.tensor([0, lk_[i]], device="cuda").int()
                alone = mfa.flash_attn_varlen_func(q[q0:q0 + lq[i]], k[k0:k0 + lk_[i]], v[k0:k0 + lk_[i]], c1q, c1k, lq[i], lk_[i], causal=causal)
                ok = ok and bool(torch.equal(alone, o[q0:q0 + lq[i]]))
            bad += not ok
            print(f"{str(dt)[6:]:9s} varlen q{lq} k{lk_} causal={int(causal)}: max {worst:.2e} {'ok' if ok else 'FAIL'}", flush=True)
# the same batches over a PAGED K/V cache (pages of 64 .. 512 keys in a permuted pool, junk in the rows past a sequence's end):
# the kernel's paged instances must reproduce the packed-K/V result of the varlen instances bit for bit
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import hip_path as hp
for dt in (torch.float16, torch.bfloat16):
    for lens in ([64], [1, 77, 300, 512, 1000], [2048, 1900], [513, 129, 640], [4096]):
        for page in (64, 128, 512):
            for causal in (False, True):
                H, Hk, B, smax = 4, 2, len(lens), max(lens)
                q = torch.randn(sum(lens), H, 128, device="cuda").to(dt)
                kd = torch.randn(B, smax, Hk, 128, device="cuda").to(dt)
                vd = torch.randn(B, smax, Hk, 128, device="cuda").to(dt)
                cu = torch.tensor([0] + lens, device="cuda").cumsum(0).int()
                kpk = torch.cat([kd[i, :n] for i, n in enumerate(lens)])
                vpk = torch.cat([vd[i, :n] for i, n in enumerate(lens)])
                want = mfa.flash_attn_varlen_func(q, kpk, vpk, cu, cu, smax, smax, causal=causal)
                assert _lib.mfa_debug_last_route() & capi.MFA_ROUTE_PREFILL64
                kp, vp, table = hp.make_paged(kd, vd, page, seed=3)
                got = mfa.flash_attn_varlen_func(q, kp, vp, cu, cu, smax, smax, causal=causal, block_table=table)
                route = _lib.mfa_debug_last_route()
                torch.cuda.synchronize()
                served = table.size(1) <= 64
                assert bool(route & capi.MFA_ROUTE_PREFILL64) == served, (route, table.shape)
                ok = bool(torch.equal(got, want)) if served else bool((got.float() - want.float()).abs().max() < 2e-2)
                bad += not ok
                print(f"{str(dt)[6:]:9s} paged q{lens} page {page} causal={int(causal)}: {'bit-equal' if served else 'general kernel'} {'ok' if ok else 'FAIL'}", flush=True)
print("FAILURES:", bad, flush=True)
if bad:
    sys.exit(1)
# forced reference-max growth: keys whose scores ramp up tile after tile (exercises the hand-over to the textbook update)
q = torch.randn(1, 512, 2, 128, device="cuda").half()
k = torch.randn(1, 512, 2, 128, device="cuda").half()
v = torch.randn(1, 512, 2, 128, device="cuda").half()
k = k + (torch.arange(512, device="cuda").view(1, 512, 1, 1) / 16.0).half() * q[:, :1].mean(dim=1, keepdim=True).sign()
for causal in (False, True):
    o = mfa.flash_attn_func(q, k, v, causal=causal)
    r = ref(q, k, v, causal)
    err = (o.float() - r).abs()
    print(f"ramp causal={int(causal)}: max {err.max().item():.2e} finite={bool(torch.isfinite(o).all())}", flush=True)
    if not (err.max().item() < 5e-3):
        sys.exit(2)

def bench(fn, n=20, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

if len(sys.argv) > 1 and sys.argv[1] == "noperf":
    sys.exit(0)
B, H, D = 48, 24, 128
for S in (1024, 2048, 4096):
    q, k, v = (torch.randn(B, S, H, D, device="cuda", dtype=torch.float16) for _ in range(3))
    for causal in (True, False):
        ms = bench(lambda: mfa.flash_attn_func(q, k, v, causal=causal), n=20 if S < 4096 else 10)
        fl = 4.0 * B * H * S * S * D * (0.5 if causal else 1.0)
        print(f"S{S} causal={int(causal)}: {ms:.3f} ms {fl / ms / 1e9:.0f} TFLOP/s", flush=True)
