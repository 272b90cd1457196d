"""Shared test plumbing.

  -m "not gpu" : oracle vs golden vectors / SDPA, host logic, C-ABI symbol + argument checks (no GPU needed)
  -m gpu       : parity tests proper — the HIP path (through the C ABI and through the Python API) against the
                 oracle, the golden fixtures and SDPA, on a real MI355X.

Tolerance (BASELINE.json north_star): |out - ref_fp32| <= atol + rtol*|ref| with atol = rtol = 1e-3, plus the
half-ulp of the OUTPUT dtype (2^-11 fp16, 2^-8 bf16) times |ref|, which no kernel that returns fp16/bf16 can
avoid.  `assert_close` below is that bar, written out.

Prefill in bf16 gets 3e-3 more absolute slack (`p_rounded=True`): the reference algorithm rounds P to the element
type before P.V (prefill.cuh:555-574; the C restatement does the same), which perturbs O by up to
2^-8 * sum(p|v|)/l — invisible in fp16 (2^-11), not in bf16.  Decode keeps P in fp32 and needs no such term.
"""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(ROOT, "mini-flash-attention_amd")
for p in (ROOT, PKG_DIR):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")
ATOL = 1e-3
RTOL = 1e-3
HALF_ULP = {torch.float16: 2.0 ** -11, torch.bfloat16: 2.0 ** -8, torch.float32: 0.0}
P_ROUND_ATOL = {torch.float16: 0.0, torch.bfloat16: 3e-3, torch.float32: 0.0}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def has_gpu():
    return torch.cuda.is_available()


def pytest_collection_modifyitems(config, items):
    if has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def assert_close(out, ref, out_dtype=None, atol=ATOL, rtol=RTOL, what="", p_rounded=False):
    """out: tensor in fp16/bf16 (or fp32); ref: fp32 reference of the same shape."""
    out_dtype = out_dtype or out.dtype
    if p_rounded:
        atol = atol + P_ROUND_ATOL[out_dtype]
    o, r = out.detach().float().cpu(), ref.detach().float().cpu()
    assert o.shape == r.shape, f"{what}: shape {tuple(o.shape)} vs {tuple(r.shape)}"
    assert torch.isfinite(o).all(), f"{what}: non-finite output"
    bound = atol + (rtol + HALF_ULP[out_dtype]) * r.abs()
    err = (o - r).abs()
    worst = (err - bound).max().item()
    assert worst <= 0, (f"{what}: max|err|={err.max().item():.3e} mean={err.mean().item():.3e}, "
                        f"exceeds atol+rtol*|ref|+half_ulp by {worst:.3e}")


def strict_report(out, ref, name, must_pass=True, why=""):
    """The literal north_star bar: |out - ref_fp32| <= 1e-3 + 1e-3*|ref| -- no half-ulp of the output type, no
    P-rounding slack.  Records max / mean error and the violating fraction under `name` in parity_r03.json (written
    beside the other run outputs: gpurun_out/ on the GPU box, copied to profiles/); asserts when must_pass."""
    import json
    o, r = out.detach().float().cpu(), ref.detach().float().cpu()
    assert o.shape == r.shape and torch.isfinite(o).all(), name
    err = (o - r).abs()
    bad = err > ATOL + RTOL * r.abs()
    rec = {"max_abs_err": float(err.max()), "mean_abs_err": float(err.mean()), "max_abs_ref": float(r.abs().max()),
           "violating_fraction": float(bad.float().mean()), "elements": int(err.numel()), "out_dtype": str(out.dtype),
           "bar": "|out - ref| <= 1e-3 + 1e-3*|ref| (strict)", "meets_bar": not bool(bad.any())}
    if why:
        rec["note"] = why
    outdir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(outdir, exist_ok=True)
    path = os.path.join(outdir, os.environ.get("MFA_PARITY_RECORD", "parity_r03.json"))  # (a forced-route child run keeps its own file)
    try:
        with open(path) as f:
            allrec = json.load(f)
    except (OSError, ValueError):
        allrec = {}
    allrec[name] = rec
    with open(path, "w") as f:
        json.dump(allrec, f, indent=1, sort_keys=True)
    if must_pass:
        assert not bad.any(), f"{name}: strict bar violated by {rec['violating_fraction']:.2e} of elements, max|err|={rec['max_abs_err']:.3e}"
    return rec


def from_bits(a, dtype):
    """uint16 numpy array (golden fixture) -> torch tensor of fp16/bf16 with the same bits."""
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int16)).view(dtype)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def mfa():
    """The product package; fails loudly if the HIP extension is not built."""
    import mini_flash_attention
    return mini_flash_attention


@pytest.fixture(scope="session")
def capi():
    """ctypes binding of libmfa_hip.so (the C ABI of include/mfa.h)."""
    import torch  # noqa: F401  (load torch's HIP runtime first so both share it)
    from mini_flash_attention import capi as c
    if os.environ.get("MFA_TEST_CAPI_LIB"):  # tests/test_sanitizers_cpu.py: the host half alone, sanitized, no kernels
        c.load(os.environ["MFA_TEST_CAPI_LIB"])
    c.load()
    return c
