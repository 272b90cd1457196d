for pr in 0 1 0 1; do echo PAIRED=$pr; MFA_PAIRED=$pr timeout -k 10 200 python tools/perf_sweep.py prefill 2>&1 | grep "prefill"; done
