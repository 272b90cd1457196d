for nw in 4 8 4 8; do echo NW=$nw; MFA_NW=$nw timeout -k 10 200 python tools/perf_sweep.py prefill --quick 2>&1 | grep "prefill"; done
