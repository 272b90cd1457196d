"""In-tree build of the native parts (no JIT cache, no hipify, no cmake):

  csrc/mfa_{prefill,decode}.hip + csrc/mfa_capi.cpp  --hipcc, gfx950-->  mini_flash_attention/libmfa_hip.so   (C ABI, include/mfa.h)
  csrc/torch_binding.cpp                             --g++, torch hdrs-->  mini_flash_attention/_C.<abi>.so    (pybind11 module)

hipcc cross-compiles for gfx950 without a GPU.  Objects are cached under build/ by source mtime.
Usage: python build.py [--force] [--no-torch]
"""
import os
import subprocess
import sys
import sysconfig
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
PKG = os.path.join(HERE, "mini_flash_attention")
BUILD = os.path.join(HERE, "build")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
HIPCC = os.path.join(ROCM, "bin", "hipcc")
ARCH = "gfx950"

HIP_FLAGS = [
    f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-fno-math-errno",
    "-Wno-unused-result", "-mllvm", "-amdgpu-early-inline-all=true",
]
if os.environ.get("MFA_EXTRA_HIPCC_FLAGS"):  # developer builds: csrc/mfa_dev.h switches (-DMFA_DEV_*), backend options
    HIP_FLAGS += os.environ["MFA_EXTRA_HIPCC_FLAGS"].split()


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout + "\n")
        raise RuntimeError(f"build step failed: {cmd[0]} ... {cmd[-1]}")
    return r.stdout


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    hs.append(os.path.join(ROOT, "include", "mfa.h"))
    hs.append(os.path.abspath(__file__))
    return hs


def torch_lib_dir():
    import torch
    return os.path.join(os.path.dirname(torch.__file__), "lib")


def build_hip_lib(force=False):
    import hashlib
    # objects are cached per flag set: a developer build (MFA_EXTRA_HIPCC_FLAGS) never leaks into the product library
    objdir = os.path.join(BUILD, hashlib.sha1(" ".join(HIP_FLAGS).encode()).hexdigest()[:10])
    os.makedirs(objdir, exist_ok=True)
    srcs = ["mfa_prefill.hip", "mfa_prefill64.hip", "mfa_decode.hip", "mfa_kvcache.hip", "mfa_capi.cpp"]
    objs, jobs = [], []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + _headers()):
            cmd = [HIPCC] + HIP_FLAGS + (["-x", "hip"] if s.endswith(".cpp") else []) + ["-c", src, "-o", obj]
            jobs.append(cmd)
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(_run, jobs))
    out = os.path.join(PKG, "libmfa_hip.so")
    tag = os.path.join(BUILD, "linked_from")  # (which flag set the library in the package was linked from)
    last = open(tag).read() if os.path.exists(tag) else ""
    if force or jobs or _stale(out, objs) or last != objdir:
        # Link against the HIP runtime by SONAME (libamdhip64.so.7).  Inside a torch process the loader
        # reuses the runtime torch already mapped (same SONAME), so streams and pointers are shared;
        # standalone it resolves through the rpath to ROCm's copy.
        rpaths = ["$ORIGIN", os.path.join(ROCM, "lib")]
        cmd = [HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", out] + objs
        cmd += ["-Wl,-soname,libmfa_hip.so"] + [f"-Wl,-rpath,{p}" for p in rpaths]
        _run(cmd)
        with open(tag, "w") as f:
            f.write(objdir)
    return out


def build_torch_ext(force=False):
    import pybind11
    import torch
    tinc = os.path.join(os.path.dirname(torch.__file__), "include")
    tlib = torch_lib_dir()
    ext = sysconfig.get_config_var("EXT_SUFFIX")
    out = os.path.join(PKG, "_C" + ext)
    src = os.path.join(CSRC, "torch_binding.cpp")
    lib = os.path.join(PKG, "libmfa_hip.so")
    if not (force or _stale(out, [src, lib] + _headers())):
        return out
    cmd = [
        "g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
        "-DTORCH_EXTENSION_NAME=_C", "-DTORCH_API_INCLUDE_EXTENSION_H", "-D_GLIBCXX_USE_CXX11_ABI=1",
        "-DUSE_ROCM", "-D__HIP_PLATFORM_AMD__=1",
        f"-I{tinc}", f"-I{os.path.join(tinc, 'torch', 'csrc', 'api', 'include')}",
        f"-I{pybind11.get_include()}", f"-I{sysconfig.get_paths()['include']}",
        f"-I{os.path.join(ROCM, 'include')}", f"-I{os.path.join(ROOT, 'include')}",
        src, "-o", out,
        f"-L{PKG}", "-lmfa_hip", f"-L{tlib}", "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch",
        "-ltorch_python",
        "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{tlib}", "-Wl,--no-as-needed",
    ]
    _run(cmd)
    return out


def build_all(force=False, with_torch=True):
    outs = [build_hip_lib(force)]
    if with_torch:
        outs.append(build_torch_ext(force))
    return outs


if __name__ == "__main__":
    for o in build_all(force="--force" in sys.argv, with_torch="--no-torch" not in sys.argv):
        print("built", os.path.relpath(o, ROOT))
