"""The in-kernel merge of key-split partials (the last split of a row to arrive runs combine_row in its own epilogue:
flash decoding and the packed-row kv-cache kernels) against the same launches with the merge as decode_combine_kernel's
own launch (MFA_FUSED_COMBINE=0, read once per process: two child processes).  Same partials, same merge function: the
outputs must agree bit for bit; each child also checks that repeated launches agree (arrival counters reset) and reports,
per case, which merge the library says it ran (mfa_debug_last_route): the first child must have merged in the kernel on
every split case, the second never -- a test that is green against itself because both children fell back is refused."""
import os
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_in_kernel_merge_equals_combine_launch(tmp_path):
    outs = []
    for flag in ("1", "0"):
        path = str(tmp_path / f"out{flag}.pt")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "dump_kvcache_outputs.py"), path],
                           env=dict(os.environ, MFA_FUSED_COMBINE=flag), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
        outs.append(torch.load(path, weights_only=True))
        routes = [int(x) for x in r.stdout.split("routes", 1)[1].split("units")[0].split()]
        units = [int(x) for x in r.stdout.split("units", 1)[1].split()]
        split = [(x, u) for x, u in zip(routes, units) if x & (8 | 16)]  # MFA_ROUTE_COMBINE_LAUNCH | MFA_ROUTE_FUSED_MERGE
        spread = lambda u: u < 64 and u % 8 != 0  # (spread_splits, csrc/mfa_launch.h: such launches' splits go out over all XCDs)
        assert len(routes) == 10 and len(split) >= 8 and 1 <= sum(spread(u) for _, u in split) <= 4, (routes, units)
        # in-kernel merge wherever it is allowed
        assert all(x & (8 | 16) == (16 if flag == "1" and not spread(u) else 8) for x, u in split), (flag, routes, units)
    assert outs[0].keys() == outs[1].keys() and len(outs[0]) == 10
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), f"{k}: in-kernel merge differs from the combine launch"


def test_in_kernel_merge_under_uneven_load_with_reused_workspaces(mfa, capi):
    """The hand-off the in-kernel merge rests on -- partials written by other workgroups, read by the last one to arrive with
    L2-served loads and no L1 invalidate -- checked the way such hand-offs fail: the SAME workspace addresses launch after
    launch (the reader's L1 may still hold the previous launch's partials), new data every launch, another stream keeping
    part of the chip busy (uneven arrival), every output word compared with the combine launch's on the same partials."""
    import hip_path as hp
    lib = capi.load()
    dev = "cuda"
    g = torch.Generator().manual_seed(11)
    B, H, Hk, Sk, D, splits = 8, 24, 8, 4096, 128, 4          # 64 rows x 4 splits = 256 workgroups: merged in the kernel
    kc = torch.randn(B, Sk, Hk, D, generator=g).to(torch.bfloat16).to(dev)
    vc = torch.randn(B, Sk, Hk, D, generator=g).to(torch.bfloat16).to(dev)
    ws = (torch.empty(splits * B * H * D, dtype=torch.float32, device=dev), torch.empty(splits * B * H, dtype=torch.float32, device=dev))
    ws2 = (torch.empty_like(ws[0]), torch.empty_like(ws[1]))
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
    stop = torch.zeros(1, device=dev)
    with torch.cuda.stream(side):                               # a few ms of matmuls on half the chip's worth of work at a time
        for _ in range(60):
            stop += (a @ a).float().mean() * 0
    for it in range(40):
        q = torch.randn(B, 1, H, D, generator=g).to(torch.bfloat16).to(dev)
        lens = torch.randint(Sk // 2, Sk + 1, (B,), generator=g).int().to(dev)
        fused = hp.decode("capi", mfa, capi, q, kc, vc, lens, num_splits=splits, ws=ws)
        assert lib.mfa_debug_last_route() == capi.MFA_ROUTE_DECODE | capi.MFA_ROUTE_FUSED_MERGE
        plain = hp.decode("capi", mfa, capi, q, kc, vc, lens, num_splits=splits, ws=ws2, counters=False)
        assert lib.mfa_debug_last_route() == capi.MFA_ROUTE_DECODE | capi.MFA_ROUTE_COMBINE_LAUNCH
        assert torch.equal(fused, plain), f"launch {it}: {(fused.float() - plain.float()).abs().max().item():.3e}"
    torch.cuda.synchronize()
