"""Where a workgroup of prefill64_kernel spends its time: shader-clock stamps of the last wave of every workgroup, second
work item (developer aid; needs a DEVELOPER build -- the stamps and the timing-only loop variants are compiled out of the
product, csrc/mfa_dev.h):
    python tools/gen_p64_stream.py --dev && MFA_EXTRA_HIPCC_FLAGS=-DMFA_DEV_P64 python mini-flash-attention_amd/build.py
    python tools/p64_timeline.py [S] [causal 0/1] [4 * variant]
    python mini-flash-attention_amd/build.py        # back to the product build"""
import os, sys
import torch
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
causal = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
B, H, D = 48, 24, 128
nwg = 8 * ((B * H + 7) // 8 // 4 + 1) * 4 * ((S + 255) // 256) + 64
dbg = torch.zeros(nwg * 16, device="cuda", dtype=torch.int64)
os.environ["MFA_P64_DBGPTR"] = str(dbg.data_ptr())
os.environ["MFA_P64_DEBUG"] = str(2 | (int(sys.argv[3]) if len(sys.argv) > 3 else 0))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mini-flash-attention_amd"))
import mini_flash_attention as mfa
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _knobs; _knobs.apply()
q, k, v = (torch.randn(B, S, H, D, device="cuda", dtype=torch.float16) for _ in range(3))
for _ in range(5):
    mfa.flash_attn_func(q, k, v, causal=causal)
dbg.zero_()
mfa.flash_attn_func(q, k, v, causal=causal)
torch.cuda.synchronize()
d = dbg.view(-1, 16).cpu()
d = d[d[:, 0] != 0]
nt = (d[:, 7] >> 32).float(); ntw = (d[:, 7] & 0xffffffff).float()
t0 = d[:, 0].min().item()
names = ["prev item done->barrier", "tile DMA issue", "init->first tiles done (X0, softmax, X1)", "loop: first steady block", "loop total",
         "epilogue"]
seg = [(0, 1), (1, 2), (2, 3), (3, 4), (3, 5), (5, 6)]
print(f"S={S} causal={causal}: {len(d)} workgroups, kernel span {(d[:, 6].max().item() - t0)} cycles (100 MHz? see below)")
for ntv in sorted(set(nt.tolist())):
    m = nt == ntv
    row = [f"nt={int(ntv):3d} n={int(m.sum()):5d}"]
    for (a, b), nm in zip(seg, names):
        x = (d[m, b] - d[m, a]).float()
        x = x[(d[m, b] != 0) & (d[m, a] != 0)]
        row.append(f"{x.mean().item() if len(x) else float('nan'):9.0f}")
    tot = (d[m, 6] - d[m, 0]).float()
    row.append(f"total {tot.mean().item():9.0f}")
    print(" ".join(row))
print("columns:", " | ".join(names))
x = (d[:, 1] - d[:, 0]).float()
print(f"  item boundary (end of previous epilogue -> barrier passed): mean {x.mean().item():7.0f} min {x.min().item():7.0f} p50 {x.median().item():7.0f} max {x.max().item():7.0f}")

m = nt == nt.max()
for i in range(1, -1, -1):
    a = (d[m, 9 + 2 * i] - d[m, 8 + 2 * i]).float(); b = ((d[m, 8 + 2 * (i - 1)] if i else d[m, 5]) - d[m, 9 + 2 * i]).float()
    print(f"  iteration nt-{i}: barrier wait {a.mean().item():7.0f}  body {b.mean().item():7.0f}")
for a, b, nm in ((13, 14, "K0' K1' V0' DMA issue"), (14, 15, "Q' DMA issue")):
    x = (d[m, b] - d[m, a]).float()
    print(f"  prefetch {nm:24s} {x.mean().item():7.0f}")
