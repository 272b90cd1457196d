"""Dense causal prefill with few (batch, head) pairs (fp16 D128): the 64-row kernel (256-row work items, one workgroup per CU)
against the general one (128-row workgroups, two per CU), steady us (developer probe; one child per kernel).
python tools/small_batch_point.py"""
import os, subprocess, sys, time
if os.environ.get("SMALL_CHILD") != "1":
    res = {}
    for flag in ("2", "0", "1"):
        out = subprocess.run([sys.executable, __file__], env=dict(os.environ, SMALL_CHILD="1", MFA_PREFILL64=flag), capture_output=True, text=True).stdout
        for ln in out.splitlines():
            if "=" in ln:
                k, ms = ln.split("=")
                res.setdefault(k, []).append(float(ms))
    for k, (a, b, c) in res.items():
        print(f"{k}: 64-row {a * 1e3:8.1f} us | general {b * 1e3:8.1f} us | general/64-row {b / a:.3f} | launcher {c * 1e3:8.1f} us")
    sys.exit(0)
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mini-flash-attention_amd"))
import mini_flash_attention as mfa
for B, H, Hk, S in ((1, 12, 12, 8192), (1, 20, 4, 4096), (1, 28, 4, 8192), (3, 12, 4, 2048), (1, 40, 8, 8192), (1, 4, 4, 4096), (1, 4, 1, 16384), (1, 6, 2, 8192), (1, 8, 8, 2048), (1, 8, 8, 4096), (1, 8, 8, 8192), (1, 32, 8, 2048), (1, 32, 8, 4096), (1, 32, 8, 8192), (1, 32, 8, 16384),
                    (2, 32, 8, 2048), (4, 32, 8, 2048), (1, 64, 8, 4096), (8, 8, 8, 1024), (4, 16, 16, 1024), (2, 24, 8, 1024), (8, 24, 8, 1024)):
    q = torch.randn(B, S, H, 128, device="cuda", dtype=torch.float16)
    k, v = (torch.randn(B, S, Hk, 128, device="cuda", dtype=torch.float16) for _ in range(2))
    for causal in (True, False):
        f = lambda: mfa.flash_attn_func(q, k, v, causal=causal)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.25:
            for _ in range(10): f()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): f()
        e1.record(); torch.cuda.synchronize()
        print(f"B{B} H{H}/{Hk} S{S} {'c' if causal else 'n'}={e0.elapsed_time(e1) / 30:.5f}", flush=True)
