"""Phase timeline of the workgroups of the general prefill kernel (prefill_fwd_kernel; developer tool, needs a build
with the stamps of csrc/mfa_dev.h compiled in; the 64-rows-per-wave kernel has tools/p64_timeline.py):

    MFA_EXTRA_HIPCC_FLAGS=-DMFA_DEV_TIMELINE python mini-flash-attention_amd/build.py
    MFA_PREFILL64=0 python tools/wg_timeline.py [S] [causal]
    python mini-flash-attention_amd/build.py        # back to the product build

The instrumented kernel variant writes, per workgroup, four 100 MHz timestamps (entry, prologue done = Q + first
K/V tile landed, tile loop done, output stores issued) plus HW_ID / XCC_ID into the LSE buffer.  This script turns
them into: time per phase, the gap between a workgroup leaving a CU and the next one starting there, and how long
each CU holds 0 / 1 / 2 workgroups.
"""
import os
import sys
from collections import defaultdict

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mini-flash-attention_amd"))
import mini_flash_attention as mfa  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
causal = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
B, H, D = 48, 24, 128
torch.manual_seed(0)
q, k, v = (torch.randn(B, S, H, D, device="cuda", dtype=torch.float16) for _ in range(3))
for _ in range(3):
    out, lse = mfa.flash_attn_func(q, k, v, causal=causal, return_softmax_lse=True)
torch.cuda.synchronize()
raw = lse.contiguous().view(-1).view(torch.int64).cpu().numpy()
raw = raw[: (raw.size // 8) * 8].reshape(-1, 8)
rec = raw[raw[:, 7] == 0x5A5A5A5A5A5A5A5A]
print(f"S={S} causal={causal}: {len(rec)} workgroups recorded")
t0, t1, t2, t3, hw, xcc, nt = (rec[:, i] for i in range(7))
tick = 0.01  # us per 100 MHz tick
base = t0.min()
print(f"kernel span (first entry -> last store issue): {(t3.max() - base) * tick:.1f} us")
pro, loop, epi = (t1 - t0) * tick, (t2 - t1) * tick, (t3 - t2) * tick
for name, a in (("prologue", pro), ("loop", loop), ("epilogue", epi), ("loop/tile", loop / np.maximum(nt, 1))):
    print(f"  {name:10s} mean {a.mean():7.2f}  p10 {np.percentile(a, 10):7.2f}  p50 {np.percentile(a, 50):7.2f}  p90 {np.percentile(a, 90):7.2f} us")
for n in sorted(set(nt.tolist())):
    m = nt == n
    print(f"  tiles={n:3d}: n={m.sum():5d}  prologue {pro[m].mean():6.2f}  loop {loop[m].mean():7.2f} ({loop[m].mean() / max(n, 1):5.2f}/tile)  epilogue {epi[m].mean():5.2f} us")

cu_id = (hw >> 8) & 0xF
sh_id = (hw >> 12) & 1
se_id = (hw >> 13) & 7
key = ((xcc & 0xF) << 12) | (se_id << 8) | (sh_id << 4) | cu_id
per_cu = defaultdict(list)
for i, kk in enumerate(key.tolist()):
    per_cu[kk].append(i)
print(f"distinct CUs seen: {len(per_cu)}; workgroups per CU min/mean/max: "
      f"{min(map(len, per_cu.values()))}/{np.mean(list(map(len, per_cu.values()))):.1f}/{max(map(len, per_cu.values()))}")
gaps, occ = [], np.zeros(4)
span_lo, span_hi = t0.min(), t3.max()
for kk, idx in per_cu.items():
    idx = sorted(idx, key=lambda i: t0[i])
    free = []  # end times of workgroups that left this CU and whose slot has not been re-used yet
    ends = []
    ev = []
    for i in idx:
        ev.append((t0[i], 1))
        ev.append((t3[i], -1))
        cand = [e for e in ends if e <= t0[i]]
        if cand and len(ends) >= 2:
            e = max(cand)
            gaps.append((t0[i] - e) * tick)
            ends.remove(e)
        ends.append(t3[i])
    ev.sort()
    n, last = 0, span_lo
    for t, d in ev:
        occ[min(n, 3)] += t - last
        last = t
        n += d
    occ[0] += span_hi - last
gaps = np.array(gaps)
print(f"gap (a workgroup's stores issued -> next workgroup entering on that CU): mean {gaps.mean():.2f}  "
      f"p10 {np.percentile(gaps, 10):.2f}  p50 {np.percentile(gaps, 50):.2f}  p90 {np.percentile(gaps, 90):.2f} us  (n={len(gaps)})")
occ /= occ.sum()
print(f"CU time holding 0/1/2/3+ workgroups: {occ[0] * 100:.1f} % / {occ[1] * 100:.1f} % / {occ[2] * 100:.1f} % / {occ[3] * 100:.1f} %")
# start skew and tail
by_cu_end = np.array([max(t3[i] for i in idx) for idx in per_cu.values()])
print(f"CU finish times relative to the last one: mean {(span_hi - by_cu_end).mean() * tick:.1f} us, max {(span_hi - by_cu_end).max() * tick:.1f} us")
first = np.array([min(t0[i] for i in idx) for idx in per_cu.values()])
print(f"CU first-entry skew: mean {(first - span_lo).mean() * tick:.1f} us, max {(first - span_lo).max() * tick:.1f} us")
