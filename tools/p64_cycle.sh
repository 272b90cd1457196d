#!/bin/bash
# The 64-rows-per-wave prefill kernel's iteration loop ON THE GPU BOX (developer aid): parity under the launch knobs, the
# textbook-update diagnostics, timing next to the general kernel and -- when abv_dev/libmfa_hip.so (a -DMFA_DEV_P64 build,
# tools/build_variant.sh) travelled along -- the cycle timeline.   tools/p64_cycle.sh <tag> [quick]
tag=$1; out=gpurun_out; mkdir -p $out
export MFA_PREFILL64=2
fail=0
for k in "" p64_grid=8 p64_grid=24,group_pairs=8 p64_grid=8,p64_no_loop=1; do
  [ "$2" = quick ] && [ "$k" = "p64_grid=24,group_pairs=8" ] && continue
  MFA_TEST_KNOBS=$k timeout -k 10 150 python tools/p64_check.py noperf > $out/${tag}_check_$k.txt 2>&1; rc=$?
  echo "check [$k] rc=$rc ok=$(grep -c ' ok$' $out/${tag}_check_$k.txt)"; grep "FAIL\|ramp\|Error\|error" $out/${tag}_check_$k.txt | head -6
  [ $rc -ne 0 ] && fail=1
done
MFA_TEST_KNOBS=p64_grid=8 timeout -k 10 100 python tools/p64_diag.py > $out/${tag}_diag.txt 2>&1; rc=$?
echo "diag rc=$rc worst=$(grep -o 'max [0-9.e+-]*' $out/${tag}_diag.txt | awk '{print $2}' | sort -g | tail -1)"
[ $rc -ne 0 ] && fail=1
[ $fail -ne 0 ] && { echo "PARITY FAILED: no timing"; exit 1; }
unset MFA_PREFILL64
timeout -k 10 200 python tools/p64_perf.py 2>&1 | grep "^S" | tee $out/${tag}_perf.txt
if [ -f abv_dev/libmfa_hip.so ]; then
  cp abv_dev/libmfa_hip.so mini-flash-attention_amd/mini_flash_attention/libmfa_hip.so
  for c in 1 0; do timeout -k 10 100 python tools/p64_timeline.py 1024 $c > $out/${tag}_tl_1024_c$c.txt 2>&1; done
  grep -v amdgpu.ids $out/${tag}_tl_1024_c1.txt | tail -10; grep -v amdgpu.ids $out/${tag}_tl_1024_c0.txt | tail -10
fi
