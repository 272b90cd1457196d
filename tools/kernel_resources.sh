#!/bin/bash
# Per-kernel register / LDS / scratch usage of one HIP source (compiles device code only, to /tmp).
#   tools/kernel_resources.sh mini-flash-attention_amd/csrc/mfa_prefill.hip [filter]
SRC=$1; FILTER=${2:-.}
OUT=/tmp/kres_$$; mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-gpu-rdc -fno-math-errno -mllvm -amdgpu-early-inline-all=true \
   -I "$(dirname "$SRC")" --cuda-device-only -S "$SRC" -o $OUT/k.s 2>$OUT/err.txt || { cat $OUT/err.txt; exit 1; }
python3 - "$OUT/k.s" "$FILTER" <<'PY'
import re,sys
txt=open(sys.argv[1]).read(); flt=sys.argv[2]
for b in txt.split('.amdhsa_kernel ')[1:]:
    name=b.split('\n')[0]
    if not re.search(flt,name): continue
    g=lambda k: re.search(r'\.amdhsa_'+k+r'\s+(\S+)', b).group(1)
    print(f"{name[:80]:80s} vgpr={g('next_free_vgpr'):>4s} accum_off={g('accum_offset'):>4s} sgpr={g('next_free_sgpr'):>3s} scratch={g('private_segment_fixed_size')}")
PY
echo "asm: $OUT/k.s"
