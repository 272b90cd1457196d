"""The head-dim-128 prefill kernel runs instruction streams that are GENERATED (tools/gen_p64_stream.py ->
csrc/mfa_prefill64_stream.inc, committed): the reviewed source is the generator, the compiled one the .inc.  These tests
tie the two together on the CPU (hipcc cross-compiles for gfx950 without a GPU):
  * regenerating gives the committed file byte for byte;
  * both element types assemble into kernels within the register file, without scratch;
  * the register ranges the streams own (the generator's VB/NV, AB/NA) are touched by NO compiler-generated instruction
    between the asm blocks, except the moves that feed / read the operands pinned to them -- the streams keep O, Q, S, l
    and m there across blocks, and nothing else enforces that hipcc leaves them alone."""
import importlib.util
import os
import re
import subprocess
import sys

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "mini-flash-attention_amd", "csrc")
GEN = os.path.join(ROOT, "tools", "gen_p64_stream.py")


def test_committed_streams_are_the_generators_output(tmp_path):
    out = str(tmp_path / "regen.inc")
    r = subprocess.run([sys.executable, GEN, "--out", out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    committed = open(os.path.join(CSRC, "mfa_prefill64_stream.inc"), "rb").read()
    assert open(out, "rb").read() == committed, "csrc/mfa_prefill64_stream.inc is stale: run python tools/gen_p64_stream.py"


@pytest.fixture(scope="module")
def p64_asm(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("p64") / "p64.s")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "-fno-math-errno", "-mllvm",
           "-amdgpu-early-inline-all=true", "-I", CSRC, "--cuda-device-only", "-S", os.path.join(CSRC, "mfa_prefill64.hip"), "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    return open(out).read()


def _kernels(asm):
    """{demangled-ish name: (body text, resource block)} of every kernel in the device assembly"""
    out = {}
    for m in re.finditer(r"^(_ZN3mfa3p6416prefill64_kernel\w+):[^\n]*\n(.*?)^\s*\.amdhsa_kernel \1\n(.*?)\.end_amdhsa_kernel", asm, flags=re.M | re.S):
        out[m.group(1)] = (m.group(2), m.group(3))
    return out


def test_both_element_types_assemble_within_the_register_file(p64_asm):
    ks = _kernels(p64_asm)
    assert len(ks) == 6 and any("4Half" in k for k in ks) and any("6BFloat" in k for k in ks), list(ks)
    for name, (_, res) in ks.items():
        g = lambda key: int(re.search(r"\.amdhsa_" + key + r"\s+(\d+)", res).group(1))
        assert g("next_free_vgpr") <= 512 and g("private_segment_fixed_size") == 0, (name, g("next_free_vgpr"))
        assert g("group_segment_fixed_size") == 0  # (dynamic LDS: the launcher asks for all 160 KiB)


def test_compiler_code_stays_out_of_the_streams_registers(p64_asm):
    spec = importlib.util.spec_from_file_location("gen_p64", GEN)
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    v_lo, v_hi, a_lo, a_hi = gen.VB, gen.VB + gen.NV, gen.AB, gen.AB + gen.NA

    def regs(line):
        found = set()
        for kind, lo, hi in re.findall(r"\b([va])\[(\d+):(\d+)\]", line):
            found |= {(kind, i) for i in range(int(lo), int(hi) + 1)}
        found |= {(kind, int(i)) for kind, i in re.findall(r"\b([va])(\d+)\b", line)}
        return found

    owned = lambda r: (r[0] == "v" and v_lo <= r[1] < v_hi) or (r[0] == "a" and a_lo <= r[1] < a_hi)
    for name, (body, _) in _kernels(p64_asm).items():
        outside = re.sub(r";;#ASMSTART.*?;;#ASMEND", "", body, flags=re.S)
        assert outside != body, "no inline-asm markers found in the device assembly"
        touching = []
        for line in outside.splitlines():
            code = line.split(";")[0].strip()
            if not code or code.endswith(":") or code.startswith("."):
                continue
            if any(owned(r) for r in regs(code)):
                touching.append(code)
        # the only owned registers compiler code may touch are the operands PINNED to their homes: the inputs of P64_SETUP
        # (LDS read addresses, DMA lane offsets, 4h) and the outputs of P64_FINAL (l, m)
        pinned = {("v", gen.VB + r) for r in ([gen.KRD(k) for k in range(8)] + [gen.VRD(d) for d in range(4)] + [gen.V_KGO, gen.V_VGO, gen.H4] +
                                              list(gen.QGO) + [gen.L(0), gen.L(1), gen.M(0), gen.M(1)])}
        stray = sorted({r for c in touching for r in regs(c) if owned(r)} - pinned)
        assert not stray and 0 < len(touching) <= 64, (name, stray, touching[:10])
