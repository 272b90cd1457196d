# usage: bash tools/abl.sh "0 32 0 32"   (interleaved rounds of ablation/variant codes, one process each)
for a in ${1:-0 1 3 4 8 7 15}; do echo ABL=$a; MFA_ABLATE=$a timeout -k 10 200 python tools/perf_sweep.py prefill --quick 2>&1 | grep "prefill fp16"; done
