"""BASELINE config 5 (paged decode bf16 B16 Skv4096 page 256 Hq24/Hkv8 D128, 4 rotating caches) through both kv-cache routes
and forced split counts (developer aid; MFA_KVCACHE_PACKED is read once per process: two children).  python tools/config5_routes.py"""
import os, subprocess, sys, time
if os.environ.get("C5_CHILD") != "1":
    for flag in ("1", "0"):
        out = subprocess.run([sys.executable, __file__], env=dict(os.environ, C5_CHILD="1", MFA_KVCACHE_PACKED=flag), capture_output=True, text=True)
        print(("packed-row kernel: " if flag == "1" else "vector kernel:     ") + " ".join(l for l in out.stdout.splitlines() if l.startswith("s")), flush=True)
    sys.exit(0)
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mini-flash-attention_amd"))
import mini_flash_attention as mfa
B, Sk, H, Hk, D, page = 16, 4096, 24, 8, 128, 256
nb = Sk // page
sets = []
for _ in range(4):
    kp, vp = (torch.randn(B * nb, page, Hk, D, device="cuda", dtype=torch.bfloat16) for _ in range(2))
    sets.append((kp, vp, torch.randperm(B * nb, device="cuda").int().view(B, nb)))
q = torch.randn(B, 1, H, D, device="cuda", dtype=torch.bfloat16)
lens = torch.full((B,), Sk, device="cuda", dtype=torch.int32)
st = {"i": 0}
for splits in (0, 1, 2, 3, 4, 6, 8):
    def run():
        kp, vp, t = sets[st["i"] % 4]; st["i"] += 1
        mfa.flash_attn_with_kvcache(q, kp, vp, cache_seqlens=lens, block_table=t, num_splits=splits)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2:
        for _ in range(20): run()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): run()
    e1.record(); torch.cuda.synchronize()
    print(f"s{splits}={e0.elapsed_time(e1) / 200 * 1e3:.1f}us")
