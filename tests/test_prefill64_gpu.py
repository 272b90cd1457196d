"""GPU parity of the 64-rows-per-wave prefill kernel (csrc/mfa_prefill64.hip) under launch geometries a default launch
does not reach on small inputs, set through the library's test hook (mfa_test_set_knob; the child tools apply
MFA_TEST_KNOBS, tools/_knobs.py): a small persistent grid (every workgroup walks many work items, ring
slots rotate across items, the snake order's odd steps), other scheduling group sizes, and the kernel driven entirely
through its per-phase blocks (p64_no_loop: the steady-state loop block never entered).  Each case is one child python
running tools/p64_check.py (11 shapes x fp16/bf16 x causal vs SDPA-fp32, plus a ramp that forces the textbook update tile
after tile) and tools/p64_diag.py (spikes per tile: which chain / tile a stream bug would hit)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def run(tool, env_extra, *args):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), *args], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    return r.stdout


@pytest.mark.parametrize("knobs", ["p64_grid=8", "p64_grid=24,group_pairs=8", "group_pairs=1", "p64_no_loop=1"])
def test_parity_under_launch_knobs(knobs):
    out = run("p64_check.py", {"MFA_TEST_KNOBS": knobs}, "noperf")
    assert "FAILURES: 0" in out and out.count("ramp causal") == 2, out[-1500:]


@pytest.mark.parametrize("knobs", ["", "p64_grid=8", "p64_grid=8,p64_no_loop=1"])
def test_textbook_update_tile_by_tile(knobs):
    out = run("p64_diag.py", {"MFA_TEST_KNOBS": knobs})
    for line in out.splitlines():
        if " max " in line:
            worst = float(line.split(" max ")[1].split()[0])
            assert worst < 2e-3, line
    assert out.count("spike tile") >= 14 and "ramp causal" in out and "multi ramp down causal" in out


def test_bit_determinism():
    out = run("p64_det.py", {})
    assert out.count(" deterministic ") == 4 and "NONDETERMINISTIC" not in out, out[-1500:]


def test_prefill_suite_with_the_64_row_kernel_forced():
    """The parity suite of the prefill entry and the long-sequence fuzz once more in a child with MFA_PREFILL64=2: every shape the 64-row kernel accepts
    (dense below the routing threshold, ragged varlen batches, paged K/V with pages >= 64 keys) goes to it instead of the
    general kernel the launcher would pick (the tests that pin the launcher's own choice skip that assertion)."""
    env = dict(os.environ, MFA_PREFILL64="2", MFA_PARITY_RECORD="parity_r03_forced64.json")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_prefill_gpu.py"),
                        os.path.join(ROOT, "tests", "test_fuzz_gpu.py") + "::test_fuzz_long_sequence_prefill_d128", "-q", "-x", "-p", "no:cacheprovider"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and " passed" in r.stdout, (r.stdout + r.stderr)[-2000:]
