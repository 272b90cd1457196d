"""One timing point per headline decode shape (config 3, config 5, README MHA Skv 1024; rotating caches, us) -- a probe for
tools/ab_libs.sh (AB_POINT=tools/ab_decode_point.py)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mini-flash-attention_amd"))
import mini_flash_attention as mfa
def point(name, B, H, Hk, S, dt, page=0, copies=3):
    q = torch.randn(B, 1, H, 128, device="cuda", dtype=dt)
    sets = []
    for _ in range(copies):
        k, v = (torch.randn(B, S, Hk, 128, device="cuda", dtype=dt) for _ in range(2))
        if page:
            nb = B * S // page
            perm = torch.randperm(nb, device="cuda")
            kp, vp = torch.empty(nb, page, Hk, 128, device="cuda", dtype=dt), torch.empty(nb, page, Hk, 128, device="cuda", dtype=dt)
            kp[perm], vp[perm] = k.view(nb, page, Hk, 128), v.view(nb, page, Hk, 128)
            sets.append((kp, vp, perm.int().view(B, S // page).contiguous()))
        else:
            sets.append((k, v, None))
    cl = torch.full((B,), S, device="cuda", dtype=torch.int32)
    it = [0]
    def f():
        k, v, t = sets[it[0] % copies]; it[0] += 1
        return mfa.flash_attn_with_kvcache(q, k, v, cache_seqlens=cl, block_table=t)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.4:
        for _ in range(10): f()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): f()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}={e0.elapsed_time(e1) * 10:.1f}")
point("config3", 24, 24, 8, 8192, torch.bfloat16, copies=2)
point("config5", 16, 24, 8, 4096, torch.bfloat16, page=256, copies=4)
point("mha1024", 24, 24, 24, 1024, torch.float16, copies=5)
point("g8", 24, 64, 8, 8192, torch.bfloat16, copies=2)
