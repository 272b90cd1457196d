"""Prefill timing of the head-dim-128 kernels over the headline shapes (fp16 B48 H24 D128), the 64-rows-per-wave kernel and the
general kernel side by side in one session (clocks differ between boxes):   python tools/p64_perf.py
(developer aid; runs itself twice, MFA_PREFILL64=1 / 0)"""
import os, subprocess, sys
if os.environ.get("P64_PERF_CHILD") != "1":
    res = {}
    for flag in ("1", "0"):
        env = dict(os.environ, P64_PERF_CHILD="1", MFA_PREFILL64=flag)
        out = subprocess.run([sys.executable, __file__], env=env, capture_output=True, text=True).stdout
        for ln in out.splitlines():
            if ln.startswith("S"):
                k, ms = ln.split(":")
                res.setdefault(k, []).append(float(ms))
    for k, (a, b) in res.items():
        S, c = (int(x[1:]) for x in k.split())
        fl = 4 * 48 * 24 * S * S * 128 / (2 if c else 1)
        print(f"{k}: p64 {a:.3f} ms {fl / a / 1e9:5.0f} TF | general {b:.3f} ms {fl / b / 1e9:5.0f} TF | ratio {b / a:.3f}")
    sys.exit(0)
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mini-flash-attention_amd"))
import mini_flash_attention as mfa
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _knobs; _knobs.apply()
B, H, D = 48, 24, 128
for S in (1024, 2048, 4096):
    q, k, v = (torch.randn(B, S, H, D, device="cuda", dtype=torch.float16) for _ in range(3))
    for causal in (True, False):
        for _ in range(10):
            mfa.flash_attn_func(q, k, v, causal=causal)
        torch.cuda.synchronize()
        n = 30 if S < 4096 else 10
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            mfa.flash_attn_func(q, k, v, causal=causal)
        e1.record()
        torch.cuda.synchronize()
        print(f"S{S} c{int(causal)}:{e0.elapsed_time(e1) / n}")
