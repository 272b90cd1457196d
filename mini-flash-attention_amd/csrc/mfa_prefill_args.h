// Kernel argument block shared by the prefill kernels (mfa_prefill.hip: 32 query rows per wave, every head dim and
// mode; mfa_prefill64.hip: 64 query rows per wave, head dim 128, dense).  Filled from mfa_forward_params
// (include/mfa.h, the mirror of the reference's ForwardParams, csrc/mfa/flash.h:8-73) by make_args().
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mfa {

struct PrefillArgs {
    const void* q;
    const void* k;
    const void* v;
    void* o;
    const int32_t* cu_q;
    const int32_t* cu_k;
    const int32_t* block_table;
    int64_t q_batch_stride, q_head_stride, q_row_stride;
    int64_t k_batch_stride, k_head_stride, k_row_stride;
    int64_t v_batch_stride, v_head_stride, v_row_stride;
    int64_t o_batch_stride, o_head_stride, o_row_stride;
    int64_t k_block_stride, v_block_stride, table_batch_stride;
    int32_t batch, heads, kv_heads, group, head_dim;
    int32_t seqlen_q, seqlen_k; // dense lengths, or max lengths when varlen
    int32_t page_size, page_shift, max_blocks;
    int32_t num_m_blocks;
    int32_t group_pairs; // (batch, head) pairs per scheduling group
    int32_t interleave_pairs; // general kernel: pairs dealt round-robin over the XCDs instead of in contiguous ranges: 1 (batch,
                              // head) pairs, 2 (batch, KV head) pairs with their query heads, 3 fewer than 8 pairs: plain order
    int32_t p64_forced;  // MFA_PREFILL64=2 (tests, probes): the 64-row kernel for everything it serves, whatever the launch size
    int32_t p64_ragged;  // 64-row kernel, varlen: the launcher found the batch ragged (mean length < 0.9 max): length-sorted schedule
    int32_t is_causal;
    float scale_log2;
    // widening beyond the reference's surface (all off when zero / null):
    const int32_t* seqlens_k; // dense or paged K cache with a per-batch valid length (kv-cache attention, Sq > 1)
    float* lse;               // natural-log LSE out: dense (B,H,Sq), varlen (H,total_q); null = not wanted
    int64_t total_q;
    int32_t has_hi, hi_off;   // keep key <= row + off + hi_off   (causal: has_hi = 1, hi_off = 0)
    int32_t has_lo, lo_off;   // keep key >= row + off + lo_off   (sliding window: lo_off = -window_left)
    int32_t seqlens_k_offset; // added to seqlens_k[b]
    int32_t bottom_right;     // off = sk - sq (flash-attn >= 2.1 alignment) instead of 0 (the reference's top-left)
    float scale;              // softmax_scale (for the LSE)
    // packed-row kv-cache attention (MQ kernels): the G query heads of a KV head x the seqlen_q query positions
    // form ONE block of rows (row = position * G + head-in-group), keys are split over workgroups
    int32_t mq_rows;          // seqlen_q * group
    int32_t mq_row_blocks;    // ceil(mq_rows / BM)
    int32_t num_splits;       // key splits; > 1: normalised partial O and LSE go to o_acc / lse_acc
    float* o_acc;             // (splits, B, Sq, H, D) fp32
    float* lse_acc;           // (splits, B, Sq, H) fp32
    int32_t* split_ctr;       // one arrival counter per (batch, KV head, row block), zero between launches: the last split
                              // to arrive merges the partials itself (no combine launch); null: the caller launches it
};

// 64-rows-per-wave kernel (mfa_prefill64.hip).  Returns -2 when the shape / mode is not one it serves (the caller
// then launches the general kernel), 0 on success, -3 on a launch error.
int launch_prefill64(PrefillArgs& a, bool is_bf16, hipStream_t stream);

} // namespace mfa
