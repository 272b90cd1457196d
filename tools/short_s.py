"""Short-sequence prefill (fp16 B48 H24 D128, S = 128 .. 512): the 64-rows-per-wave kernel against the general one with 8 and
with 4 waves per workgroup, steady us (developer aid; MFA_PREFILL64 is read once per process: one child per variant).
python tools/short_s.py"""
import os, subprocess, sys, time
if os.environ.get("SHORT_S_CHILD") != "1":
    res = {}
    for flag, knobs in (("2", ""), ("0", "nw8=1"), ("0", "nw8=0")):
        out = subprocess.run([sys.executable, __file__], env=dict(os.environ, SHORT_S_CHILD="1", MFA_PREFILL64=flag, MFA_TEST_KNOBS=knobs), capture_output=True, text=True).stdout
        for ln in out.splitlines():
            if ln.startswith("S"):
                k, ms = ln.split("=")
                res.setdefault(k, []).append(float(ms))
    for k, (a, b, c) in res.items():
        print(f"{k}: p64 {a * 1e3:7.1f} us | general, 8 waves {b * 1e3:7.1f} us | general, 4 waves {c * 1e3:7.1f} us")
    sys.exit(0)
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mini-flash-attention_amd"))
import mini_flash_attention as mfa
import _knobs
_knobs.apply()
B, H, D = 48, 24, 128
for S in (128, 192, 256, 320, 384, 512):
    q, k, v = (torch.randn(B, S, H, D, device="cuda", dtype=torch.float16) for _ in range(3))
    for causal in (True, False):
        f = lambda: mfa.flash_attn_func(q, k, v, causal=causal)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.3:
            for _ in range(20): f()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): f()
        e1.record(); torch.cuda.synchronize()
        print(f"S{S}{'c' if causal else 'n'}={e0.elapsed_time(e1) / 100:.5f}")
