# effective shader clock of the prefill kernel: GRBM_GUI_ACTIVE/8/duration, for the library as built (to compare ablated
# builds of the general kernel, rebuild between calls: MFA_EXTRA_HIPCC_FLAGS=-DMFA_DEV_ABL_MASK=<bits>, csrc/mfa_dev.h)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp; cd /tmp
for a in ${1:-0}; do
  OUT=$ROOT/gpurun_out/clk_$a; mkdir -p $OUT
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/tools/run_shape.py prefill 4096 0 4 > $OUT/log.txt 2>&1
  python3 - $OUT $a <<'PY'
import csv,glob,sys
out,a=sys.argv[1],sys.argv[2]
rows=[r for r in csv.DictReader(open(glob.glob(out+'/*/*_counter_collection.csv')[0])) if 'prefill' in r['Kernel_Name']]
by={}
for r in rows:
    by.setdefault(r['Dispatch_Id'],{})[r['Counter_Name']]=float(r['Counter_Value']); by[r['Dispatch_Id']]['dur']=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
for d,v in list(by.items())[1:]:
    print(f"ABL={a} dur={v['dur']/1e6:.3f} ms  GRBM/8={v['GRBM_GUI_ACTIVE']/8/1e6:.2f} Mcyc  clock={v['GRBM_GUI_ACTIVE']/8/v['dur']:.3f} GHz  wave_cycles={v['SQ_WAVE_CYCLES']/1e9:.3f}G")
PY
done
