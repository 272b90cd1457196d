"""Packed variable-length prefill (bf16 H24/8 D128 causal): the 64-rows-per-wave kernel's varlen instances against the general
kernel, steady ms, on an even batch and on ragged ones (developer probe; MFA_PREFILL64 is read once per process: two children;
=2 forces the 64-row kernel past the launcher's mean-length test).   python tools/varlen_point.py"""
import os, subprocess, sys, time
if os.environ.get("VARLEN_POINT_CHILD") != "1":
    res = {}
    for flag in ("2", "0", "1"):
        out = subprocess.run([sys.executable, __file__], env=dict(os.environ, VARLEN_POINT_CHILD="1", MFA_PREFILL64=flag), capture_output=True, text=True)
        for ln in out.stdout.splitlines():
            if "=" in ln:
                k, ms = ln.rsplit("=", 1)
                res.setdefault(k, []).append(ms)
        if out.returncode:
            print(out.stderr[-1500:])
    for k, v in res.items():
        a, b, c = (x.split() for x in v)
        print(f"{k}: 64-row kernel {float(a[0]):.4f} ms | general {float(b[0]):.4f} ms | ratio {float(b[0]) / float(a[0]):.3f} | launcher's choice: {'64-row' if c[1] == '1' else 'general'} {float(c[0]):.4f} ms")
    sys.exit(0)
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mini-flash-attention_amd"))
import mini_flash_attention as mfa
from mini_flash_attention import capi
lib = capi.load()
torch.manual_seed(0)
H, Hk, D = 24, 8, 128
g = torch.Generator().manual_seed(1)
cases = {
    "even 16 x 2048": [2048] * 16,
    "even 48 x 1024": [1024] * 48,
    "1024..2048 x 24": torch.randint(1024, 2049, (24,), generator=g).tolist(),
    "512..4096 x 16": torch.randint(512, 4097, (16,), generator=g).tolist(),
    "128..4096 log x 32": (128 * 2 ** (5 * torch.rand(32, generator=g))).int().tolist(),
    "one 8192 + 31 x 256": [8192] + [256] * 31,
    "300..1500 x 100": torch.randint(300, 1501, (100,), generator=g).tolist(),
    "200 x 512..1024": torch.randint(512, 1025, (200,), generator=g).tolist(),
}
for name, lens in cases.items():
    T = sum(lens)
    q = torch.randn(T, H, D, device="cuda", dtype=torch.bfloat16)
    k, v = (torch.randn(T, Hk, D, device="cuda", dtype=torch.bfloat16) for _ in range(2))
    cu = torch.tensor([0] + lens, device="cuda").cumsum(0).int()
    f = lambda: mfa.flash_attn_varlen_func(q, k, v, cu, cu, max(lens), max(lens), causal=True)
    f()
    p64 = 1 if lib.mfa_debug_last_route() & capi.MFA_ROUTE_PREFILL64 else 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.4:
        for _ in range(5): f()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): f()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}={e0.elapsed_time(e1) / 30:.5f} {p64}", flush=True)
