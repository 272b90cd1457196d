"""Where a workgroup of prefill64_kernel spends its time: shader-clock stamps of the last wave of every workgroup, second
work item (developer aid; needs a DEVELOPER build -- the stamps and the timing-only loop variants are compiled out of the
product, csrc/mfa_dev.h):
    tools/build_variant.sh abv_dev -DMFA_DEV_P64;  on the GPU box: cp abv_dev/libmfa_hip.so mini-flash-attention_amd/mini_flash_attention/
    python tools/p64_timeline.py [S] [causal 0/1] [4 * variant] [stamping wave 0..3]
Stamps (the wave that runs through the item boundary): 0 previous item's loop left | 1 item set up | 2 -> 3 barrier of the
item's first iteration | 8 tile requests issued | 9 joint block done (P.V of the old item's last tile, textbook softmax of the
new item's first, scores of its second) | 10 old item's epilogue done | 11 next item's Q requested | 4 loop block left at
iteration gl-2 | 12, 13 calls of the loop block for iterations gl-1, gl | 5 loop left."""
import os, sys
import torch
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
causal = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
B, H, D = 48, 24, 128
nwg = 8 * ((B * H + 7) // 8 // 4 + 1) * 4 * ((S + 255) // 256) + 64
dbg = torch.zeros(nwg * 64, device="cuda", dtype=torch.int64)
os.environ["MFA_P64_DBGPTR"] = str(dbg.data_ptr())
WAVE = int(sys.argv[4]) if len(sys.argv) > 4 else 3
os.environ["MFA_P64_DEBUG"] = str(2 | (int(sys.argv[3]) if len(sys.argv) > 3 else 0) | (WAVE << 8))
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
d = dbg.view(-1, 64).cpu()
d = d[d[:, 0] != 0]
nt = (d[:, 7] >> 32).float()
segs = [("boundary: prev loop left -> item set up", 0, 1), ("first barrier wait", 2, 3), ("tile requests", 3, 8),
        ("joint block / drain + epilogue", 8, 10), ("next item's Q requests", 10, 11), ("first iteration in all", 1, 11),
        ("loop block (last call) -> loop left", 11, 5), ("item in all (loop left -> loop left)", 0, 5)]
print(f"S={S} causal={causal}: {len(d)} workgroups stamped")
for ntv in sorted(set(nt.tolist())):
    m = nt == ntv
    print(f"nt={int(ntv):3d} n={int(m.sum()):5d}")
    for nm, a, b in segs:
        ok = m & (d[:, a] != 0) & (d[:, b] != 0)
        x = (d[ok, b] - d[ok, a]).float()
        if len(x):
            print(f"    {nm:44s} mean {x.mean().item():8.0f}  min {x.min().item():8.0f}  max {x.max().item():8.0f}")
    # barriers this wave passed outside the loop block (iteration k of the item: stamp 16 + k) and its calls of the loop block
    ev = []
    for k in range(48):
        ok = m & (d[:, 16 + k] != 0) & (d[:, 0] != 0)
        if ok.any():
            ev.append((f"it{k}" if k < 32 else f"loop@{2 * (k - 32)}", (d[ok, 16 + k] - d[ok, 0]).float().mean().item()))
    ev.sort(key=lambda e: e[1])
    print(f"    wave {WAVE}: barrier passed / loop block called, cycles after the previous item's loop was left:  " + "  ".join(f"{n}:{t:.0f}" for n, t in ev))
