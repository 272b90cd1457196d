"""AddressSanitizer + UndefinedBehaviorSanitizer over the CPU-side native code (the reference ships a
compute-sanitizer recipe for its kernels; GPU sanitizers are not available on this pool, so the CPU builds carry it):
  * oracle/mfa_oracle.c — the restatement every parity test leans on — runs its golden-vector and SDPA checks
    (tests/test_oracle_cpu.py) from a sanitized build;
  * the HOST half of the C ABI (csrc/mfa_capi.cpp: validation, split heuristics, workspace sizing, kv-cache plan)
    runs its CPU tests (tests/test_capi_cpu.py) from a sanitized build linked against test-only launcher stubs
    (tests/sanitize/host_stubs.cpp).
The sanitized objects are loaded into a child python started with the sanitizer runtimes preloaded; any report makes
the child exit non-zero (halt_on_error, -fno-sanitize-recover).  CPU only — never run on the GPU box."""
import os
import shutil
import subprocess
import sys

import pytest

from conftest import ROOT

SAN = ["-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]
CSRC = os.path.join(ROOT, "mini-flash-attention_amd", "csrc")


def _runtime(name):
    p = subprocess.check_output(["gcc", f"-print-file-name={name}"], text=True).strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.fixture(scope="module")
def sanitized(tmp_path_factory):
    if shutil.which("gcc") is None or _runtime("libasan.so") is None or _runtime("libubsan.so") is None:
        pytest.skip("gcc sanitizer runtimes not installed")
    out = tmp_path_factory.mktemp("san")
    oracle_so, capi_so = str(out / "libmfa_oracle_san.so"), str(out / "libmfa_capi_host_san.so")
    subprocess.check_call(["gcc", *SAN, "-fPIC", "-shared", os.path.join(ROOT, "oracle", "mfa_oracle.c"), "-o", oracle_so, "-lm"])
    subprocess.check_call(["g++", *SAN, "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           f"-I{CSRC}", os.path.join(CSRC, "mfa_capi.cpp"), os.path.join(ROOT, "tests", "sanitize", "host_stubs.cpp"),
                           "-o", capi_so])
    env = dict(os.environ, LD_PRELOAD=f"{_runtime('libasan.so')}:{_runtime('libubsan.so')}",
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0:exitcode=97",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1:exitcode=98",
               MFA_TEST_ORACLE_LIB=oracle_so, MFA_TEST_CAPI_LIB=capi_so, OMP_NUM_THREADS="1")
    return env


def _run(env, args):
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", *args], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, f"sanitized run failed (exit {r.returncode}):\n{tail}"
    assert "passed" in r.stdout and "ERROR: AddressSanitizer" not in tail and "runtime error:" not in tail, tail


def test_oracle_restatement_under_asan_ubsan(sanitized):
    _run(sanitized, ["tests/test_oracle_cpu.py"])


def test_capi_host_half_under_asan_ubsan(sanitized):
    _run(sanitized, ["tests/test_capi_cpu.py", "-k",
                     "set_scale or num_splits_semantics or workspace_bytes or argument_validation or kvcache_plan or struct_layout or exports_every"])
