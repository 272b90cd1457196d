"""No kernel of the library may use scratch memory: tools/kernel_resources.sh (hipcc -S for gfx950, no GPU needed) over
every .hip source, failing on any `private_segment_fixed_size != 0` (a spill, a dynamically indexed private array, or
the emergency stack slot the backend reserves at the scalar-register limit)."""
import glob
import os
import re
import subprocess
from concurrent.futures import ThreadPoolExecutor

from conftest import ROOT

SOURCES = sorted(glob.glob(os.path.join(ROOT, "mini-flash-attention_amd", "csrc", "*.hip")))


def _resources(src):
    out = subprocess.run(["bash", os.path.join(ROOT, "tools", "kernel_resources.sh"), src], capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = re.findall(r"^(\S+)\s+vgpr=\s*(\d+)\s+accum_off=\s*(\d+)\s+sgpr=\s*(\d+)\s+scratch=(\d+)", out.stdout, flags=re.M)
    return [(name, int(v), int(s), int(sc)) for name, v, _, s, sc in rows]


def test_no_kernel_uses_scratch():
    assert len(SOURCES) >= 4, SOURCES
    with ThreadPoolExecutor(max_workers=4) as ex:
        per_file = list(ex.map(_resources, SOURCES))
    total = 0
    bad = []
    for src, rows in zip(SOURCES, per_file):
        assert rows, f"no kernels found in {src}"
        total += len(rows)
        bad += [f"{os.path.basename(src)}: {name} scratch={sc}" for name, _, _, sc in rows if sc != 0]
        assert all(v <= 512 for _, v, _, _ in rows)
    assert not bad, "kernels with a private segment:\n" + "\n".join(bad)
    assert total >= 100  # (the instance table: prefill x head dims x modes, decode x group tiles, ...)
