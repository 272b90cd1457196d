import os, sys
import torch
os.environ.setdefault("MFA_PREFILL64", "2")  # every shape the 64-row kernel serves goes to it
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mini-flash-attention_amd"))
import mini_flash_attention as mfa
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _knobs; _knobs.apply()
import torch.nn.functional as F
from torch.nn.attention import SDPBackend, sdpa_kernel
def ref(q, k, v, causal):
    qf, kf, vf = (t.float().transpose(1, 2) for t in (q, k, v))
    with sdpa_kernel(SDPBackend.MATH):
        return F.scaled_dot_product_attention(qf, kf, vf, is_causal=causal).transpose(1, 2)
torch.manual_seed(1)
for (dt, B, Sq, Sk, H, causal) in ((torch.bfloat16, 1, 64, 64, 1, False), (torch.float16, 3, 129, 64, 2, False), (torch.float16, 1, 512, 512, 2, True),
                                   (torch.float16, 1, 256, 256, 2, False)):
    q = torch.randn(B, Sq, H, 128, device="cuda").to(dt); k = torch.randn(B, Sk, H, 128, device="cuda").to(dt); v = torch.randn(B, Sk, H, 128, device="cuda").to(dt)
    r = ref(q, k, v, causal)
    outs = []
    for it in range(12):
        if it % 3 == 1:  # disturb: run another shape in between
            mfa.flash_attn_func(torch.randn(2, 700, 2, 128, device="cuda").to(dt), torch.randn(2, 700, 2, 128, device="cuda").to(dt), torch.randn(2, 700, 2, 128, device="cuda").to(dt), causal=True)
        o = mfa.flash_attn_func(q, k, v, causal=causal)
        torch.cuda.synchronize()
        outs.append(o.float())
    errs = [(o - r).abs().max().item() for o in outs]
    same = all(torch.equal(outs[0], o) for o in outs)
    nbad = [(( (o - r).abs() > 2e-2).sum().item()) for o in outs]
    print(str(dt)[6:], B, Sq, Sk, H, causal, "deterministic" if same else "NONDETERMINISTIC", " ".join(f"{e:.1e}" for e in errs), nbad, flush=True)
    if max(nbad) > 0:
        o = outs[nbad.index(max(nbad))]
        bad = ((o - r).abs() > 2e-2).nonzero()
        print("  bad idx sample (b, row, head, col):", bad[:12].tolist(), flush=True)
