#!/bin/bash
# Builds a DEVELOPER variant of libmfa_hip.so into <dir>/libmfa_hip.so (abv*/ is git-ignored but travels with gpurun) and
# leaves the product library in the package:   tools/build_variant.sh abv_dev -DMFA_DEV_DECODE_AB [-DMFA_DEV_P64 ...]
# On the GPU box: cp <dir>/libmfa_hip.so mini-flash-attention_amd/mini_flash_attention/libmfa_hip.so && <tool>
set -e
cd "$(dirname "$0")/.."
dir=$1; shift
case " $* " in *MFA_DEV_P64*) python tools/gen_p64_stream.py --dev >/dev/null;; esac
MFA_EXTRA_HIPCC_FLAGS="$*" python mini-flash-attention_amd/build.py --no-torch
mkdir -p "$dir"
cp mini-flash-attention_amd/mini_flash_attention/libmfa_hip.so "$dir/libmfa_hip.so"
python mini-flash-attention_amd/build.py --no-torch
echo "variant [$*] -> $dir/libmfa_hip.so; product library restored"
