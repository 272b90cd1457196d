// Host-side launchers behind the C ABI (include/mfa.h).  One translation unit per kernel family.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/mfa.h"

namespace mfa {

// Prefill / varlen / paged prefill (replaces run_mha_prefill, reference csrc/mfa/flash.cu:11-34).
int launch_prefill(const mfa_forward_params& p, hipStream_t stream);

// Decode + optional combine (replaces run_mha_decode, reference csrc/mfa/flash.cu:36-71).
int launch_decode(const mfa_forward_params& p, hipStream_t stream);

// KV-cache append (no reference counterpart: see include/mfa.h).
int launch_kvcache_append(const mfa_kvcache_append_params& p, hipStream_t stream);

} // namespace mfa
