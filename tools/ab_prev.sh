# A/B: previous commit's build (ab_prev/) vs the working tree, interleaved, same box
for r in 1 2; do
 echo PREV; MFA_PKG_DIR=$PWD/ab_prev/mini-flash-attention_amd timeout -k 10 200 python tools/perf_sweep.py ${1:-prefill} --quick 2>&1 | grep "prefill\|decode"
 echo CUR; timeout -k 10 200 python tools/perf_sweep.py ${1:-prefill} --quick 2>&1 | grep "prefill\|decode"
done
