# same-box timing of ablated builds of the general prefill kernel (csrc/mfa_dev.h bits; results are wrong by construction):
#   bash tools/abl.sh "0 1 4 8"      builds each mask into the package in turn, times it, then restores the product build
for a in ${1:-0 1 3 4 8 7 15}; do
  echo ABL=$a
  MFA_EXTRA_HIPCC_FLAGS="-DMFA_DEV_ABL_MASK=$a" python mini-flash-attention_amd/build.py > /dev/null || exit 1
  MFA_PREFILL64=0 timeout -k 10 200 python tools/perf_sweep.py prefill --quick 2>&1 | grep "prefill fp16"
done
python mini-flash-attention_amd/build.py > /dev/null
