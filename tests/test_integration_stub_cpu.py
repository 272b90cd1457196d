"""INTEGRATION.md §1 shows the file a maintainer of the reference adds (mfa::ForwardParams -> mfa_forward_params, field by
field).  It cannot be compiled here -- the reference's flash.h includes <cuda_runtime.h>, and stand-ins for absent headers
are not written -- so it is checked as text instead: every member it reads from the reference struct is declared in
/root/reference/csrc/mfa/flash.h (read where it lies, as text), every member of that struct is either copied or named
below as deliberately unused, and every member it writes exists in include/mfa.h.  Skipped where the reference is absent
(the GPU box)."""
import os
import re

import pytest

from conftest import ROOT

REF = "/root/reference/csrc/mfa/flash.h"
pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="needs the reference checkout")


def _members(struct_text):
    names = []
    for line in struct_text.splitlines():
        line = line.split("//")[0].strip()
        m = re.match(r"^[\w:<>\s\*]+?[\s\*]+(?:__restrict__\s+)?(\w+)\s*;$", line)
        if m and not line.startswith(("using", "struct", "void run_")):
            names.append(m.group(1))
    return names


def test_stub_reads_and_writes_only_declared_members():
    ref = open(REF).read()
    ref_struct = ref[ref.index("struct ForwardParams"):ref.index("};", ref.index("struct ForwardParams"))]
    ref_members = set(_members(ref_struct))
    assert {"q_ptr", "softmax_scale_log2", "block_table", "seqlens_k", "oaccum_ptr", "num_splits"} <= ref_members, sorted(ref_members)
    ours = open(os.path.join(ROOT, "include", "mfa.h")).read()
    our_struct = ours[ours.index("typedef struct mfa_forward_params"):ours.index("} mfa_forward_params;")]
    our_members = set(re.findall(r"(\w+)\s*(?:,|;)", re.sub(r"/\*.*?\*/", "", our_struct, flags=re.S)))
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    stub = doc[doc.index("static mfa_forward_params to_c("):doc.index("return p;")]
    read, written = set(re.findall(r"\bs\.(\w+)", stub)), set(re.findall(r"\bp\.(\w+)\s*=", stub))
    assert read and read <= ref_members, sorted(read - ref_members)
    assert written <= our_members, sorted(written - our_members)
    # what the stub leaves alone, and why: the maxima are passed as the lengths already (api.cpp:236); the rest is derived
    unused = {"max_seqlen_q", "max_seqlen_k", "total_q", "total_k"} & ref_members
    assert ref_members - read <= unused, sorted(ref_members - read - unused)
    # the two entry points it defines are the reference's (flash.h:76-77)
    for fn in ("run_flash_attention_forward", "run_flash_attention_with_kv_cache"):
        assert re.search(r"void\s+" + fn + r"\s*\(\s*ForwardParams\s*&", ref) and fn in doc
