// Host-side launchers behind the C ABI (include/mfa.h).  One translation unit per kernel family.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/mfa.h"

namespace mfa {

// Prefill / varlen / paged prefill (replaces run_mha_prefill, reference csrc/mfa/flash.cu:11-34).
int launch_prefill(const mfa_forward_params& p, hipStream_t stream);

// Decode + optional combine (replaces run_mha_decode, reference csrc/mfa/flash.cu:36-71).
int launch_decode(const mfa_forward_params& p, hipStream_t stream);

// Packed-row kv-cache attention, seqlen_q >= 1 (MQ instances of the prefill kernel, mfa_prefill.hip) with p.num_splits
// key splits and, when > 1, the combine below.  Returns -2 when no instance exists for the head dim.
int launch_kvcache_packed(const mfa_forward_params& p, hipStream_t stream);
int launch_decode_combine(const mfa_forward_params& p, hipStream_t stream);

// Arrival counters for the in-kernel merge of key splits (mfa_prefill.hip): n zeroed int32 for this (device, stream), or
// null when the merge has to stay a separate launch (see there).  The kernels leave them zeroed.
int32_t* split_counters(hipStream_t stream, size_t n);

// KV-cache append (no reference counterpart: see include/mfa.h).
int launch_kvcache_append(const mfa_kvcache_append_params& p, hipStream_t stream);

} // namespace mfa
