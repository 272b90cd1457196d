#!/bin/bash
# rocprofv3 passes over bench.py on the GPU box (run through gpurun from the repo root):
#   1. --kernel-trace --stats          -> per-kernel average duration
#   2. --pmc FETCH_SIZE                -> HBM read bytes  (own pass: 3 of 4 TCC slots)
#   3. --pmc WRITE_SIZE                -> HBM write bytes (own pass)
#   4. --pmc SQ_* (busy/wave cycles)   -> MFMA busy, LDS bank conflicts
# Summaries are written under gpurun_out/<tag>/ ; copy what should be judged into profiles/.
set -u
TAG=${1:-prof}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
ARGS="$ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sweep"
# the --stats average is over EVERY launch, warm-up included (cold clocks, first-touch): enough timed steps that the
# steady state bench.py reports dominates it
ARGS_STATS="$ROOT/bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-sweep"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 $ARGS_STATS > "$OUT/stats.log" 2>&1 || exit 1
[ "${2:-}" = "stats-only" ] && exit 0
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python3 $ARGS > "$OUT/pmc_fetch.log" 2>&1 || exit 2
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python3 $ARGS > "$OUT/pmc_write.log" 2>&1 || exit 3
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_sq" -- python3 $ARGS > "$OUT/pmc_sq.log" 2>&1 || exit 4
find "$OUT" -name "*.csv" | head -40
