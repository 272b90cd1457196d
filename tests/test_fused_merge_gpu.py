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
        routes = [int(x) for x in r.stdout.split("routes", 1)[1].split()]
        split = [x for x in routes if x & (8 | 16)]  # MFA_ROUTE_COMBINE_LAUNCH | MFA_ROUTE_FUSED_MERGE
        assert len(routes) == 10 and len(split) >= 8, routes
        want = 16 if flag == "1" else 8
        assert all(x & (8 | 16) == want for x in split), (flag, routes)
    assert outs[0].keys() == outs[1].keys() and len(outs[0]) == 10
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), f"{k}: in-kernel merge differs from the combine launch"
