# wave-slot occupancy of the prefill kernel: SQ_WAVE_CYCLES*4 / (GRBM_GUI_ACTIVE/8 * CUs * 8 waves)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp; cd /tmp
S=${1:-1024}; C=${2:-1}
for g in ${3:-1 4}; do
  export MFA_TEST_KNOBS=group_pairs=$g
  OUT=$ROOT/gpurun_out/occ_$g; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/tools/run_shape.py prefill $S $C 4 > $OUT/log.txt 2>&1
  python3 - $OUT $g <<'PY'
import csv,glob,sys
out,a=sys.argv[1],sys.argv[2]
rows=[r for r in csv.DictReader(open(glob.glob(out+'/*/*_counter_collection.csv')[0])) if 'prefill' in r['Kernel_Name']]
by={}
for r in rows:
    by.setdefault(r['Dispatch_Id'],{})[r['Counter_Name']]=float(r['Counter_Value']); by[r['Dispatch_Id']]['dur']=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
for d,v in list(by.items())[1:]:
    cyc=v['GRBM_GUI_ACTIVE']/8
    print(f"GP={a} dur={v['dur']/1e3:.1f} us  clock={cyc/v['dur']:.3f} GHz  occupancy={v['SQ_WAVE_CYCLES']*4/(cyc*256*8):.3f}  waves={v['SQ_WAVES']:.0f}")
PY
done
