# A/B of alternative builds (abv1, abv2, ... directories holding a full package build) against the working tree
for r in 1 2; do
 echo CUR; timeout -k 10 200 python tools/perf_sweep.py prefill --quick 2>&1 | grep "prefill fp16" | grep -v "S 4096 causal=1"
 for d in abv*; do echo $d; MFA_PKG_DIR=$PWD/$d/mini-flash-attention_amd timeout -k 10 200 python tools/perf_sweep.py prefill --quick 2>&1 | grep "prefill fp16" | grep -v "S 4096 causal=1"; done
done
