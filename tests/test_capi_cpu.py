"""CPU-only checks of the C-ABI library (include/mfa.h): it loads without a GPU, exports every declared
symbol, its struct layout matches the ctypes mirror, and argument validation / split heuristics behave as the
reference's host layer does (csrc/mfa/api.cpp).  No kernel is launched here."""
import ctypes
import os
import re
import subprocess

import pytest
import torch

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "mfa.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mfa_[a-z_0-9]+)\s*\(", src)))


def test_header_compiles_as_plain_c(tmp_path):
    c = tmp_path / "t.c"
    c.write_text('#include "mfa.h"\nint main(void){ mfa_forward_params p; (void)p; return sizeof(p) > 0 ? 0 : 1; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", f"-I{ROOT}/include", "-c", str(c), "-o", str(tmp_path / "t.o")])


def test_library_exports_every_declared_symbol(capi):
    lib = capi.load()
    names = declared_functions()
    assert {"mfa_run_flash_attention_forward", "mfa_run_flash_attention_with_kv_cache", "mfa_num_splits_heuristic"} <= set(names)
    for n in names:
        assert hasattr(lib, n), f"libmfa_hip.so does not export {n}"
    assert lib.mfa_abi_version() == 4
    assert b"gfx950" in lib.mfa_version()


def test_struct_layout_matches_ctypes_mirror(capi):
    assert capi.load().mfa_forward_params_sizeof() == ctypes.sizeof(capi.ForwardParams)
    assert capi.load().mfa_kvcache_append_params_sizeof() == ctypes.sizeof(capi.KvAppendParams)


def test_only_one_hip_runtime_is_mapped(capi):
    """libmfa_hip.so must share the HIP runtime torch already loaded (same SONAME), not bring a second one."""
    capi.load()
    mapped = {l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l}
    assert len(mapped) == 1, mapped


def test_set_scale(capi):
    p = capi.ForwardParams()
    p.head_dim, p.heads, p.kv_heads = 128, 24, 8
    capi.load().mfa_forward_params_set_scale(ctypes.byref(p))
    assert abs(p.softmax_scale - 128 ** -0.5) < 1e-7
    assert abs(p.softmax_scale_log2 - 128 ** -0.5 * 1.4426950408889634) < 1e-7
    assert p.kv_group_size == 3


def test_num_splits_semantics(capi):
    """Argument semantics of the reference (api.cpp:305-327): <1 = auto, explicit values clamp to the number of
    64-key tiles; the auto value itself is re-derived for one workgroup per (batch, KV head)."""
    h = capi.load().mfa_num_splits_heuristic
    assert h(1, 24, 8, 8192, 256) == 1
    assert h(4, 24, 8, 8192, 256) == 4
    assert h(1000, 2, 2, 640, 256) == 10          # clamp to ceil(640/64)
    assert h(3, 2, 2, 64, 256) == 1               # one tile cannot be split
    assert h(0, 4096, 8, 8192, 256) == 1          # plenty of workgroups already
    assert h(0, 24, 8, 8192, 256) == 1            # BASELINE config 3: 192 (batch, kv head) pairs fill 3/4 of 256 CUs
    assert h(0, 4, 8, 8192, 256) in (6, 8)        # 32 pairs -> 192..256 workgroups
    assert h(0, 24, 24, 8192, 256) == 4           # MHA: 576 pairs = 2.25 per CU -> 4 splits = 9 per CU exactly
    assert h(0, 16, 8, 8192, 256) == 2            # 128 pairs -> 256 workgroups
    assert h(0, 1, 8, 512, 256) <= 2              # never below 4 tiles of 64 keys per split
    assert h(0, 24, 8, 8192, 0) >= 1              # num_cus = 0: query the device, fall back to 256 without one
    assert h(0, 1, 1, 100000, 256) <= 128
    assert h(1000, 1, 1, 100000, 256) == 128      # explicit requests are capped at 128 too (combine kernel limit)


def test_workspace_bytes(capi):
    o, l = ctypes.c_size_t(), ctypes.c_size_t()
    capi.load().mfa_decode_workspace_bytes(4, 24, 24, 128, ctypes.byref(o), ctypes.byref(l))
    assert (o.value, l.value) == (4 * 24 * 24 * 128 * 4, 4 * 24 * 24 * 4)
    capi.load().mfa_decode_workspace_bytes(1, 24, 24, 128, ctypes.byref(o), ctypes.byref(l))
    assert (o.value, l.value) == (0, 0)


def _params(capi, oracle, **over):
    q = torch.zeros(1, 64, 2, 64, dtype=torch.float16)
    p = oracle.fill_params(q, q, q, torch.empty_like(q))
    for k, v in over.items():
        setattr(p, k, v)
    return p


def test_argument_validation_without_launch(capi, oracle):
    lib = capi.load()
    fwd, dec = lib.mfa_run_flash_attention_forward, lib.mfa_run_flash_attention_with_kv_cache
    assert fwd(None, None) == capi.MFA_ERR_INVALID_ARGUMENT
    cases = [
        (dict(q_ptr=0), capi.MFA_ERR_INVALID_ARGUMENT, "non-NULL"),
        (dict(head_dim=512), capi.MFA_ERR_INVALID_ARGUMENT, "less than or equal to 256"),
        (dict(heads=8, kv_heads=3), capi.MFA_ERR_INVALID_ARGUMENT, "divisible"),
        (dict(head_dim=60), capi.MFA_ERR_UNSUPPORTED, "multiple of 8"),
        (dict(head_dim=40), capi.MFA_ERR_UNSUPPORTED, "prefill supports head_dim"),
        (dict(q_row_stride=130), capi.MFA_ERR_INVALID_ARGUMENT, "multiple of 8"),
        (dict(cu_seqlens_q=64), capi.MFA_ERR_INVALID_ARGUMENT, "together"),
    ]
    for over, code, msg in cases:
        p = _params(capi, oracle, **over)
        assert fwd(ctypes.byref(p), None) == code, over
        assert msg in capi.last_error(), (over, capi.last_error())
    p = _params(capi, oracle, q_ptr=_params(capi, oracle).q_ptr + 2)
    assert fwd(ctypes.byref(p), None) == capi.MFA_ERR_INVALID_ARGUMENT and "16-byte" in capi.last_error()
    p = _params(capi, oracle, seqlen_q=0)
    assert dec(ctypes.byref(p), None) == capi.MFA_ERR_INVALID_ARGUMENT and "seqlen_q must be >= 1" in capi.last_error()
    p = _params(capi, oracle, seqlen_q=4, cu_seqlens_q=64, cu_seqlens_k=64)
    assert dec(ctypes.byref(p), None) == capi.MFA_ERR_INVALID_ARGUMENT and "packed sequences" in capi.last_error()
    p = _params(capi, oracle, seqlen_q=1, num_splits=4)
    assert dec(ctypes.byref(p), None) == capi.MFA_ERR_WORKSPACE
    # empty problems succeed without touching the device
    assert fwd(ctypes.byref(_params(capi, oracle, batch=0)), None) == capi.MFA_OK
    assert dec(ctypes.byref(_params(capi, oracle, seqlen_q=1, batch=0)), None) == capi.MFA_OK


def test_python_api_rejects_what_the_reference_rejects(mfa):
    """reference csrc/mfa/api.cpp:125-162: RuntimeError for CPU tensors / fp32 / bad head ratios; the Python
    wrapper asserts seqlen_q == 1 for the kv-cache call (interface.py:116)."""
    q = torch.zeros(1, 4, 2, 32, dtype=torch.float16)
    with pytest.raises(RuntimeError, match="must be on CUDA"):
        mfa.flash_attn_func(q, q, q)
    with pytest.raises(RuntimeError, match="fp16 and bf16"):
        mfa.flash_attn_func(q.float(), q.float(), q.float())
    with pytest.raises(RuntimeError, match="same dtype"):
        mfa.flash_attn_func(q, q.bfloat16(), q)
    # seqlen_q > 1 on the kv-cache call is a superset here (the reference asserts seqlen_q == 1, interface.py:116);
    # CPU tensors are still rejected
    with pytest.raises(RuntimeError, match="must be on CUDA"):
        mfa.flash_attn_with_kvcache(q, q, q)
    assert mfa.__version__ == "0.1.0"
    assert set(mfa.__all__) == {"flash_attn_func", "flash_attn_varlen_func", "flash_attn_with_kvcache"}


def test_extension_abi_is_positional_like_the_reference(mfa):
    """reference csrc/api.cpp:6-8 defines the three functions without py::arg names."""
    import mini_flash_attention._C as C
    for name in ("mini_flash_attention_forward", "mini_flash_attention_varlen_forward", "mini_flash_attention_with_kvcache",
                 "forward_ex", "varlen_forward_ex", "kvcache_ex"):  # the last three are opt-in supersets
        assert hasattr(C, name)
    q = torch.zeros(1, 4, 2, 32, dtype=torch.float16)
    with pytest.raises(TypeError):
        C.mini_flash_attention_forward(q=q, k=q, v=q, out=None, is_causal=False, window_size_left=-1, window_size_right=-1)


def test_product_never_imports_the_oracle():
    """The oracle is a checker, never a fallback: nothing under mini-flash-attention_amd/ may reference it."""
    pkg = os.path.join(ROOT, "mini-flash-attention_amd")
    for dp, _, fs in os.walk(pkg):
        if "build" in dp.split(os.sep):
            continue
        for f in fs:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle" not in txt.lower() or f == "capi.py" and "oracle" not in txt.lower(), os.path.join(dp, f)


def test_kvcache_plan_without_device(capi, oracle):
    """mfa_kvcache_plan is host arithmetic: routes, split counts and workspace sizes can be checked without a GPU
    (num_cus given, so nothing queries the device)."""
    lib = capi.load()
    s, ob, lb = ctypes.c_int(), ctypes.c_size_t(), ctypes.c_size_t()

    def plan(**kw):
        p = _params(capi, oracle, **kw)
        p.num_cus = 256
        assert lib.mfa_kvcache_plan(ctypes.byref(p), ctypes.byref(s), ctypes.byref(ob), ctypes.byref(lb)) == capi.MFA_OK
        return s.value, ob.value, lb.value

    # flash decoding with a small group: exactly the decode heuristic and its workspace formula
    n, o, l = plan(batch=24, seqlen_q=1, heads=24, kv_heads=8, seqlen_k=8192, head_dim=128, num_splits=0)
    assert n == lib.mfa_num_splits_heuristic(0, 24, 8, 8192, 256)
    assert (o, l) == ((n * 24 * 24 * 128 * 4, n * 24 * 24 * 4) if n > 1 else (0, 0))
    # a few query tokens: packed kernel; few (batch, KV head) pairs on 512 workgroup slots must be split
    n, o, l = plan(batch=4, seqlen_q=8, heads=32, kv_heads=8, seqlen_k=65536, head_dim=128, num_splits=0)
    assert 8 <= n <= 128 and (o, l) == (n * 4 * 8 * 32 * 128 * 4, n * 4 * 8 * 32 * 4)
    # plenty of pairs: no split, no workspace
    assert plan(batch=256, seqlen_q=2, heads=32, kv_heads=8, seqlen_k=512, head_dim=128, num_splits=0) == (1, 0, 0)
    # explicit requests are honoured up to the number of 64-key tiles
    assert plan(batch=2, seqlen_q=4, heads=32, kv_heads=8, seqlen_k=256, head_dim=128, num_splits=64)[0] == 4
    # long query blocks run the per-head prefill kernel unsplit
    assert plan(batch=2, seqlen_q=1024, heads=32, kv_heads=8, seqlen_k=4096, head_dim=128, num_splits=8) == (1, 0, 0)
    p = _params(capi, oracle, heads=8, kv_heads=3)
    assert lib.mfa_kvcache_plan(ctypes.byref(p), None, None, None) == capi.MFA_ERR_INVALID_ARGUMENT
    # arrival counters of the in-kernel split merge: one per (batch, KV head[, head chunk / block of 128 packed rows]) while the
    # split launch is at most 512 workgroups; 0 = the library keeps the merge as its own launch (or nothing is split)
    cnt = lambda **kw: lib.mfa_kvcache_counter_count(ctypes.byref(_params(capi, oracle, **kw)))
    assert cnt(batch=4, seqlen_q=1, heads=24, kv_heads=8, seqlen_k=8192, head_dim=128, num_splits=8) == 32
    assert cnt(batch=4, seqlen_q=1, heads=24, kv_heads=8, seqlen_k=8192, head_dim=128, num_splits=1) == 0
    assert cnt(batch=24, seqlen_q=1, heads=24, kv_heads=8, seqlen_k=8192, head_dim=128, num_splits=4) == 0   # 768 workgroups
    assert cnt(batch=16, seqlen_q=1, heads=24, kv_heads=8, seqlen_k=4096, head_dim=128, num_splits=4) == 128  # vector kernel
    assert cnt(batch=2, seqlen_q=40, heads=32, kv_heads=4, seqlen_k=4096, head_dim=128, num_splits=4) == 2 * 4 * 3  # packed rows
    assert cnt(batch=2, seqlen_q=1024, heads=32, kv_heads=8, seqlen_k=4096, head_dim=128, num_splits=1) == 0


def test_test_hook_and_init_without_device(capi):
    """mfa_test_set_knob takes the six documented names only; mfa_init needs a device (a negative code and a message here, on a
    box without one: it must not crash or report success); mfa_debug_last_route starts at 0."""
    lib = capi.load()
    lib.mfa_test_set_knob.argtypes = [ctypes.c_char_p, ctypes.c_int]
    for name in (b"p64_grid", b"group_pairs", b"p64_no_loop", b"nw8", b"mq_stream", b"decode_gt_max"):
        assert lib.mfa_test_set_knob(name, 0 if name != b"mq_stream" else -1) == capi.MFA_OK
    assert lib.mfa_test_set_knob(b"no_such_knob", 1) == capi.MFA_ERR_INVALID_ARGUMENT and "no_such_knob" in capi.last_error()
    assert lib.mfa_test_set_knob(None, 1) == capi.MFA_ERR_INVALID_ARGUMENT
    import torch
    if not torch.cuda.is_available():
        assert lib.mfa_init(-1) < 0 and lib.mfa_init(0) <= 0
    assert lib.mfa_debug_last_route() == 0 or torch.cuda.is_available()
