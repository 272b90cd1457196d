// Host-side launchers behind the C ABI (include/mfa.h).  One translation unit per kernel family.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>

#include "../../include/mfa.h"

namespace mfa {

// Prefill / varlen / paged prefill (replaces run_mha_prefill, reference csrc/mfa/flash.cu:11-34).
int launch_prefill(const mfa_forward_params& p, hipStream_t stream, bool* used_prefill64 = nullptr);

// Decode + optional combine (replaces run_mha_decode, reference csrc/mfa/flash.cu:36-71).  *merged_in_kernel (may be
// null): whether the split merge ran inside the split kernel (false: unsplit, or decode_combine_kernel launched behind).
int launch_decode(const mfa_forward_params& p, hipStream_t stream, bool* merged_in_kernel);

// Packed-row kv-cache attention, seqlen_q >= 1 (MQ instances of the prefill kernel, mfa_prefill.hip) with p.num_splits
// key splits and, when > 1, the combine below.  Returns -2 when no instance exists for the head dim.
int launch_kvcache_packed(const mfa_forward_params& p, hipStream_t stream, bool* merged_in_kernel);
int launch_decode_combine(const mfa_forward_params& p, hipStream_t stream);

// In-kernel merge of key splits (mfa_prefill.hip).  xcd_premise_probe: mfa_init()'s check that workgroup ids equal mod 8
// share an XCD (1 / 0 / <0 on a HIP error; cached per device; allocates and synchronises -- never on a launch path).
// pick_split_counters: the caller's counters (mfa_forward_params::split_counters) when the merge of this launch should
// and may run in the split kernel -- n counters needed, `workgroups` in the split launch, `partial_bytes` of fp32
// partials -- else null (the launcher then runs decode_combine_kernel behind the split kernel).
int xcd_premise_probe(int device);
// units: the (batch, KV head[, head chunk]) rows the launch deals over the XCDs.  A row's key splits share an XCD (the in-kernel
// merge needs that), rows r, r + 8, ... share one: with a row count that is no multiple of 8 some XCDs carry one row more than
// others -- 9 rows: XCD 0 two, the rest one; bf16 Hq4 Hkv1 Skv32768, 9 sequences 50 us where 8 take 30, 17 take 84 where 16 take
// 48 -- and with fewer than 8 rows some carry none.  For such small launches the splits go out in plain order over all XCDs and
// the merge is its own launch.  (Kernels and launchers ask this one function.)
__host__ __device__ inline bool spread_splits(int64_t units, int splits) { return splits > 1 && units < 64 && (units & 7) != 0; }
bool fused_merge_pays(int64_t units, int64_t workgroups, int64_t pbytes);
int32_t* pick_split_counters(const mfa_forward_params& p, size_t n, int64_t units, int64_t workgroups, int64_t pbytes);
#ifdef MFA_DEV_DECODE_AB // (developer A/B builds: no size gate, tools/ab_decode_map.py measures both sides of it)
constexpr int64_t kFusedMergeMaxWorkgroups = 1 << 30;
constexpr int64_t kFusedMergeMaxPartialBytes = 1ll << 40;
#else
constexpr int64_t kFusedMergeMaxWorkgroups = 512; // (two per CU: one round of the split kernel)
constexpr int64_t kFusedMergeMaxPartialBytes = 8 << 20;
#endif
inline int64_t partial_bytes(const mfa_forward_params& p) { // (S, B, Sq, H, D) + (S, B, Sq, H) fp32
    return p.num_splits > 1 ? 4ll * p.num_splits * p.batch * p.seqlen_q * p.heads * (p.head_dim + 1) : 0;
}

// Test hooks (mfa_test_set_knob, include/mfa.h): launch geometry overrides that the parity tests use to drive paths a
// default launch does not reach on small inputs.  Zero / negative = the library's own choice.  Not tuning knobs: the
// three documented environment switches (MFA_PREFILL64, MFA_FUSED_COMBINE, MFA_KVCACHE_PACKED) are those.
struct TestKnobs {
    std::atomic<int> p64_grid{0};       // prefill64: persistent grid size (a multiple of 8)
    std::atomic<int> group_pairs{0};    // prefill kernels: (batch, head) pairs per scheduling group
    std::atomic<int> p64_no_loop{0};    // prefill64: every iteration through the per-phase blocks, never the loop block
    std::atomic<int> nw8{0};            // general prefill kernel, head dim 128: 8-wave workgroups
    std::atomic<int> mq_stream{-1};     // packed kv-cache kernel: non-temporal K/V policy forced off (0) / on (1)
    std::atomic<int> decode_gt_max{0};  // vector decode: largest group tile
};
extern TestKnobs g_knobs;

// KV-cache append (no reference counterpart: see include/mfa.h).
int launch_kvcache_append(const mfa_kvcache_append_params& p, hipStream_t stream);

} // namespace mfa
