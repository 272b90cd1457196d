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
torch.manual_seed(0)
S = 512
def run(name, q, k, v, causal=False):
    o = mfa.flash_attn_func(q, k, v, causal=causal)
    r = ref(q, k, v, causal)
    err = (o.float() - r).abs()[0]          # (S, H, D)
    per = err.amax(dim=(1, 2)).view(-1, 32).amax(dim=1)   # per 32-row block
    print(name, "max", f"{err.max().item():.2e}", "per 32-row block:", " ".join(f"{x:.1e}" for x in per.tolist()), flush=True)
q = torch.randn(1, S, 1, 128, device="cuda").half()
k = torch.randn(1, S, 1, 128, device="cuda").half()
v = torch.randn(1, S, 1, 128, device="cuda").half()
run("plain", q, k, v)
# spike: keys of ONE tile t get a big boost along q-mean direction for all rows
d = torch.ones(128, device="cuda").half()
qq = q.clone(); qq[..., :] = (q.float() * 0.2 + 1.0).half()      # all rows have positive projection on d
for t in (0, 1, 2, 3, 4, 5, 6, 7):
    kk = k.clone(); kk[0, 64 * t: 64 * t + 64] += 1.5 * d
    run(f"spike tile {t}", qq, kk, v)
kk = k.clone(); kk[0] += (torch.arange(S, device="cuda").view(S, 1, 1) / 40.0).half() * d
run("ramp", qq, kk, v)
run("ramp causal", qq, kk, v, True)
# several (batch, head) pairs: with a small persistent grid (MFA_TEST_KNOBS=p64_grid=8) every workgroup walks a few work items
# and the textbook update of an item's LAST iteration (tile 0: the keys are walked downwards) runs while the Q registers
# already hold the next item's rows
B, H = 2, 8
q = torch.randn(B, S, H, 128, device="cuda").half()
k = torch.randn(B, S, H, 128, device="cuda").half()
v = torch.randn(B, S, H, 128, device="cuda").half()
qq = (q.float() * 0.2 + 1.0).half()
def run_multi(name, q, k, v, causal=False):
    o = mfa.flash_attn_func(q, k, v, causal=causal)
    err = (o.float() - ref(q, k, v, causal)).abs()
    print(name, "max", f"{err.max().item():.2e}", "per (batch, head):", " ".join(f"{x:.1e}" for x in err.amax(dim=(1, 3)).flatten().tolist()), flush=True)
for t in (0, 1, 7):
    kk = k.clone(); kk[:, 64 * t: 64 * t + 64] += 1.5 * d
    run_multi(f"multi spike tile {t}", qq, kk, v)
    run_multi(f"multi spike tile {t} causal", qq, kk, v, True)
kk = k.clone(); kk += (torch.arange(S, device="cuda").view(1, S, 1, 1) / 40.0).half() * d
run_multi("multi ramp up", qq, kk, v)
kk = k.clone(); kk += ((S - 1 - torch.arange(S, device="cuda")).view(1, S, 1, 1) / 40.0).half() * d
run_multi("multi ramp down", qq, kk, v)
run_multi("multi ramp down causal", qq, kk, v, True)
