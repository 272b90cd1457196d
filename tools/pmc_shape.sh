#!/bin/bash
# SQ-counter passes over one shape:  tools/pmc_shape.sh <tag> <run_shape args...>
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/p1" -- python3 $ROOT/tools/run_shape.py "$@" > "$OUT/p1.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/p2" -- python3 $ROOT/tools/run_shape.py "$@" > "$OUT/p2.log" 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d "$OUT/p3" -- python3 $ROOT/tools/run_shape.py "$@" > "$OUT/p3.log" 2>&1
python3 - "$OUT" <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
agg=collections.defaultdict(list)
for f in glob.glob(out+'/p*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'mfa::' in r['Kernel_Name']:
            agg[(r['Kernel_Name'].split('(')[0][-60:], r['Counter_Name'])].append(float(r['Counter_Value']))
for (k,c),v in sorted(agg.items()):
    print(f"{k:60s} {c:28s} {sum(v)/len(v):16.1f}")
PY
