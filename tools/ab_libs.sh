#!/bin/bash
# Same-box A/B of whole libraries (developer aid): [ROUNDS=n] tools/ab_libs.sh <dirA> <dirB> ... -- each <dir>/libmfa_hip.so is copied
# over the package's library in turn and tools/ab_point.py (or $AB_POINT) timed in a fresh process; "product" = the library the box arrived with.
pkg=mini-flash-attention_amd/mini_flash_attention
cp $pkg/libmfa_hip.so /tmp/product_libmfa_hip.so
for r in $(seq 1 ${ROUNDS:-3}); do
  for d in "$@"; do
    if [ "$d" = product ]; then cp /tmp/product_libmfa_hip.so $pkg/libmfa_hip.so; else cp $d/libmfa_hip.so $pkg/libmfa_hip.so; fi
    echo "$d: $(timeout -k 10 120 python ${AB_POINT:-tools/ab_point.py} 2>&1 | grep -v amdgpu | tr '\n' ' ')"
  done
done
cp /tmp/product_libmfa_hip.so $pkg/libmfa_hip.so
