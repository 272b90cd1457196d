/*
 * mfa.h — C ABI of the MI355X (gfx950) FlashAttention-2 forward hot path.
 *
 * This is the drop-in boundary.  Every entry point takes plain pointers, sizes and a raw
 * hipStream_t (as void*); there is no torch / ATen type anywhere in this file.  The torch
 * extension `mini_flash_attention._C` (csrc/torch_binding.cpp) is one client, the ctypes
 * binding used by tests/ is another, and INTEGRATION.md shows the stub a maintainer of the
 * reference would add to its csrc/mfa/api.cpp.
 *
 * What each declaration replaces in the reference (w4096/mini-flash-attention):
 *
 *   struct mfa_forward_params                <- struct mfa::ForwardParams      csrc/mfa/flash.h:8-73
 *   mfa_run_flash_attention_forward()        <- run_flash_attention_forward()  csrc/mfa/flash.h:76, flash.cu:74-82
 *   mfa_run_flash_attention_with_kv_cache()  <- run_flash_attention_with_kv_cache()
 *                                                                              csrc/mfa/flash.h:77, flash.cu:85-93
 *   mfa_forward_params_set_scale()           <- forward_params_init() scale part  csrc/mfa/api.cpp:99-100
 *   mfa_num_splits_heuristic()               <- num_splits_heuristic() + clamp in forward_params_set_split_kv()
 *                                                                              csrc/mfa/api.cpp:269-302, 305-327
 *   mfa_decode_workspace_bytes()             <- workspace sizing in forward_params_set_split_kv()
 *                                                                              csrc/mfa/api.cpp:332-337
 *
 * Conventions (same as the reference): all strides are in ELEMENTS of the tensor's dtype;
 * q/k/v/o have a unit last-dimension stride; causal masking is TOP-LEFT aligned (key index > query
 * index is masked, csrc/mfa/prefill.cuh:416-419); inputs are fp16 or bf16, accumulation is fp32.
 *
 * All launches are asynchronous on `stream`; inputs are borrowed until the enqueued kernels
 * finish.  No LAUNCH entry point (mfa_run_*, mfa_kvcache_append) allocates, frees or synchronises: they are
 * hipGraph-capturable and every buffer they touch is the caller's.  The one entry that does touch the device on
 * its own is the optional mfa_init() below (a probe launch + a blocking copy; never called implicitly).
 * Return value: 0 on success, a negative MFA_ERR_* code otherwise; mfa_last_error() gives the
 * message for the calling thread.
 */
#ifndef MFA_H_
#define MFA_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MFA_ABI_VERSION 4

enum {
    MFA_OK = 0,
    MFA_ERR_INVALID_ARGUMENT = -1, /* shape / stride / alignment / dtype precondition violated */
    MFA_ERR_UNSUPPORTED = -2,      /* head_dim or group size outside the compiled set           */
    MFA_ERR_LAUNCH = -3,           /* HIP reported a launch error                               */
    MFA_ERR_WORKSPACE = -4         /* split decode requested without oaccum / lseaccum buffers  */
};

/* Carries the fields of mfa::ForwardParams (csrc/mfa/flash.h:8-73) under the same names and in the same order,
 * as a plain-C layout of its own: bool -> int32_t, int / size_t strides -> int64_t, void* -> typed pointers, and
 * max_seqlen_q / max_seqlen_k folded into seqlen_q / seqlen_k (api.cpp:236 passes the maxima as the lengths).  It is NOT
 * layout-compatible with the C++ struct: a host fills it field by field (INTEGRATION.md shows the copy).  Nine
 * additive fields follow the reference's (max_blocks_per_seq .. reserved), all "off" when zero. */
typedef struct mfa_forward_params {
    const void* q_ptr; /* (B,Sq,H,D) or varlen (total_q,H,D)                                   */
    const void* k_ptr; /* (B,Sk,Hkv,D), varlen (total_k,Hkv,D) or paged (nblk,page,Hkv,D)     */
    const void* v_ptr;
    void* o_ptr; /* same layout as q                                                          */

    int64_t q_batch_stride, q_head_stride, q_row_stride;
    int64_t k_batch_stride, k_head_stride, k_row_stride;
    int64_t v_batch_stride, v_head_stride, v_row_stride;
    int64_t o_batch_stride, o_head_stride, o_row_stride;

    int32_t is_causal;
    int32_t window_size_left;  /* accepted, normalised, unused (api.cpp:88-96)                 */
    int32_t window_size_right;

    int32_t heads;
    int32_t kv_heads;
    int32_t kv_group_size; /* heads / kv_heads                                                 */

    int32_t batch;
    int32_t seqlen_q; /* dense: Sq; varlen: max_seqlen_q                                       */
    int32_t seqlen_k; /* dense: Sk; varlen: max_seqlen_k; paged decode: max_blocks*page        */

    int32_t head_dim;
    float softmax_scale;      /* 1/sqrt(D)                                                     */
    float softmax_scale_log2; /* softmax_scale * log2(e)                                       */

    int32_t is_bf16; /* 0 = fp16, 1 = bf16                                                     */

    const int32_t* cu_seqlens_q; /* (B+1) or NULL                                              */
    const int32_t* cu_seqlens_k; /* (B+1) or NULL                                              */

    const int32_t* block_table; /* (B, max_blocks) or NULL                                     */
    int64_t block_table_batch_stride;
    int32_t page_block_size;
    int64_t k_cache_block_stride; /* elements between pages                                    */
    int64_t v_cache_block_stride;

    const int32_t* seqlens_k; /* kv-cache: (B) valid cache length, NULL = seqlen_k for every b (decode; also
                                 honoured by the forward entry when cu_seqlens_k is NULL: Sq > 1 on a cache)   */

    int32_t num_splits; /* decode: <1 = choose (mfa_num_splits_heuristic), 1 = no split        */

    float* softmax_lse_ptr;      /* natural-log LSE out, may be NULL. decode: (B,H); prefill: dense (B,H,Sq),
                                    varlen (H,total_q) (the reference computes it for decode only and drops it) */
    float* softmax_lseaccum_ptr; /* decode split: (S,B,H) fp32 workspace                       */
    float* oaccum_ptr;           /* decode split: (S,B,H,D) fp32 workspace                     */

    /* additive fields (all "off" when zero, so a zero-initialised struct behaves like the reference) */
    int32_t max_blocks_per_seq; /* paged: columns of block_table (bounds the table reads)      */
    int32_t num_cus;            /* 0 = query the device                                        */
    int32_t mask_bottom_right;  /* causal / window aligned to the LAST key (offset sk - sq, flash-attn >= 2.1)
                                   instead of the reference's top-left; used by kv-cache attention with Sq > 1 */
    int32_t use_local_window;   /* 1: apply local_window_left/right (sliding-window attention); the reference's
                                   window_size_* fields above stay accepted-and-ignored, as upstream         */
    int32_t local_window_left;  /* keys >= row + off - left  (-1 = unbounded)                  */
    int32_t local_window_right; /* keys <= row + off + right (-1 = unbounded)                  */
    int64_t total_q;            /* varlen: rows of q (layout of the LSE output; 0 = not given, allowed without an LSE.
                                   The launcher also reads the batch's mean length off it: an even batch of long
                                   sequences, head dim 128, takes the 64-rows-per-wave kernel)                */
    int32_t seqlens_k_offset;   /* added to every seqlens_k[b] (keys just appended by mfa_kvcache_append)     */
    int32_t reserved;
    /* kv-cache entry with num_splits > 1, optional: arrival counters for the IN-KERNEL merge of the key splits (the last
     * split of a row to arrive merges the partials in its own epilogue instead of a second launch).  A caller-owned int32
     * buffer of at least mfa_kvcache_counter_count(p) entries, all ZERO before the first launch that uses it; the kernels
     * leave it zero, so it can be reused by every later launch ON THE SAME STREAM (launches that may run concurrently
     * need buffers of their own) and must outlive any hipGraph that captured a launch with it.  NULL, too short, or
     * mfa_init() not called / failed for the device: the merge is decode_combine_kernel's own launch (the reference's
     * structure, flash.cu:60-70).  Premise of the in-kernel merge, verified by mfa_init(): workgroups whose ids are equal
     * mod 8 run on one XCD (the dispatcher's round robin), so the partials of a row meet in one L2; on a stream created
     * with a CU mask pass NULL. */
    int32_t* split_counters;
    int64_t split_counters_len; /* entries */
} mfa_forward_params;

/* KV-cache append: copy new K/V rows (B, Sn, Hkv, D) into the cache at positions seqlens_k[b] .. +Sn-1 (dense
 * (B,Sk,Hkv,D) cache or paged (num_blocks,page,Hkv,D) + block_table).  Not in the reference: its docstring promises
 * it (mini_flash_attention/interface.py:110-111) and its tests do it in Python (tests/test_flash_decoding.py:574-597);
 * semantics are flash-attn's `flash_attn_with_kvcache(k=, v=)`.  Rows that would land past the cache capacity are
 * dropped.  Strides in elements. */
typedef struct mfa_kvcache_append_params {
    const void* k_new; const void* v_new;   /* (B, Sn, Hkv, D) */
    void* k_cache; void* v_cache;
    int64_t kn_batch_stride, kn_row_stride, kn_head_stride;
    int64_t vn_batch_stride, vn_row_stride, vn_head_stride;
    int64_t kc_batch_stride, kc_row_stride, kc_head_stride; /* dense: batch stride; paged: block stride */
    int64_t vc_batch_stride, vc_row_stride, vc_head_stride;
    const int32_t* seqlens_k;   /* (B) position of the first new row; NULL = 0                                */
    const int32_t* block_table; /* (B, max_blocks) or NULL                                                    */
    int64_t block_table_batch_stride;
    int32_t batch, seqlen_new, kv_heads, head_dim;
    int32_t seqlen_k;           /* dense: cache rows per batch element; paged: max_blocks * page_block_size   */
    int32_t page_block_size, max_blocks_per_seq;
    int32_t is_bf16;            /* (only the element size matters: 2 bytes either way)                        */
} mfa_kvcache_append_params;

/* ABI / build identification. */
int mfa_abi_version(void);
const char* mfa_version(void);
const char* mfa_last_error(void);
size_t mfa_forward_params_sizeof(void); /* sizeof(mfa_forward_params) the library was built with */
size_t mfa_kvcache_append_params_sizeof(void);

/* softmax_scale = 1/sqrt(head_dim), softmax_scale_log2 = softmax_scale*log2(e); kv_group_size. */
void mfa_forward_params_set_scale(mfa_forward_params* p);

/* Prefill / varlen / paged-prefill forward: O = softmax(scale*Q K^T + mask) V. */
int mfa_run_flash_attention_forward(const mfa_forward_params* p, void* hip_stream);

/* Attention of (B, seqlen_q, H, D) queries over a K/V cache with optional split-KV + LSE combine.
 * seqlen_q == 1 is the reference's flash decoding (run_flash_attention_with_kv_cache, flash.h:77).  seqlen_q > 1
 * (not in the reference: speculative / chunked decoding) treats the queries as the LAST seqlen_q positions:
 * is_causal and the local window align to the last valid key (flash-attn >= 2.1), whatever mask_bottom_right says.
 * seqlens_k (+ seqlens_k_offset), block_table and softmax_lse_ptr ((B,H,seqlen_q)) are honoured.
 * p->num_splits must already be resolved (>= 1, from mfa_kvcache_plan); workspaces must be present when it is > 1. */
int mfa_run_flash_attention_with_kv_cache(const mfa_forward_params* p, void* hip_stream);

/* Optional, once per device (-1 = current), NOT capturable: checks on the hardware that workgroup ids equal mod 8 share
 * an XCD (HW_REG_XCC_ID), the premise of the in-kernel split merge.  Allocates 4 bytes, launches a probe on the null
 * stream, copies the answer back (blocking) and frees.  Returns 1 when the premise holds (the merge may be used on
 * this device from now on), 0 when it does not (split_counters is ignored), < 0 on a HIP error.  Thread-safe; repeated
 * calls return the cached answer. */
int mfa_init(int device);

/* int32 entries the kv-cache entry would use from p->split_counters for this problem (p->num_splits resolved by
 * mfa_kvcache_plan): 0 when the launch is unsplit or when the library keeps the merge as its own launch for this size
 * (large launches, where the merge launch is cheaper than the winners' cache invalidations; small launches whose row count is
 * no multiple of 8, whose splits are spread over all XCDs).  Never more than
 * MFA_SPLIT_COUNTERS_MAX, so a caller may allocate that many once. */
size_t mfa_kvcache_counter_count(const mfa_forward_params* p);
#define MFA_SPLIT_COUNTERS_MAX 65536

/* Which kernels the calling thread's last successful mfa_run_flash_attention_with_kv_cache() or
 * mfa_run_flash_attention_forward() launched: a test / tracing aid (the reference has no counterpart).  MFA_ROUTE_* bits. */
enum {
    MFA_ROUTE_DECODE = 1,       /* vector flash-decoding kernel (mfa_decode.hip)                         */
    MFA_ROUTE_PACKED = 2,       /* packed-row MFMA kernel (MQ instances, mfa_prefill.hip)                */
    MFA_ROUTE_PREFILL = 4,      /* per-head prefill kernel (the forward entry; the kv-cache entry for seqlen_q > 1) */
    MFA_ROUTE_COMBINE_LAUNCH = 8, /* split merge as decode_combine_kernel's own launch                  */
    MFA_ROUTE_FUSED_MERGE = 16, /* split merge inside the split kernel (split_counters used)             */
    MFA_ROUTE_PREFILL64 = 32    /* with MFA_ROUTE_PREFILL: the 64-rows-per-wave kernel (mfa_prefill64.hip)  */
};
int mfa_debug_last_route(void);

/* What the kv-cache entry wants for this problem (shape fields, num_cus and num_splits of *p are read; num_splits
 * < 1 = choose): the key-split count to put into p->num_splits and the bytes of the two fp32 workspaces
 * (oaccum: (S,B,Sq,H,D), lseaccum: (S,B,Sq,H); both 0 when S == 1).  For seqlen_q == 1 and a GQA group <= 4 this
 * is mfa_num_splits_heuristic + mfa_decode_workspace_bytes. */
int mfa_kvcache_plan(const mfa_forward_params* p, int* num_splits, size_t* oaccum_bytes, size_t* lse_bytes);

/* Append new K/V rows to the cache (see mfa_kvcache_append_params). */
int mfa_kvcache_append(const mfa_kvcache_append_params* p, void* hip_stream);

/* Split count the decode path would choose for this problem on a device with `num_cus` CUs
 * (0 = query the current device).  `requested` < 1 means "auto"; an explicit request is only
 * clamped to the number of 64-key tiles, as the reference does (api.cpp:325-327). */
int mfa_num_splits_heuristic(int requested, int batch, int kv_heads, int seqlen_k, int num_cus);

/* Bytes of fp32 workspace needed for `num_splits`: *oaccum_bytes for (S,B,H,D) and *lse_bytes
 * for (S,B,H).  Both are 0 when num_splits <= 1. */
void mfa_decode_workspace_bytes(int num_splits, int batch, int heads, int head_dim,
                                size_t* oaccum_bytes, size_t* lse_bytes);

/* TEST HOOK, not part of the drop-in surface: overrides a launch-geometry choice for the whole process so that the parity
 * tests can drive paths a default launch does not reach on small inputs ("p64_grid", "group_pairs", "p64_no_loop",
 * "nw8", "mq_stream", "decode_gt_max"; 0, or -1 for mq_stream, restores the library's choice).  Results never depend on
 * a knob.  The library reads exactly three environment variables, once per process, as documented tuning switches:
 *   MFA_PREFILL64=0|1|2      head-dim-128 dense prefill: 0 = always the general kernel, 2 = the 64-rows-per-wave kernel for
 *                            every shape it serves (default 1: not for keys that fit three tiles);
 *   MFA_FUSED_COMBINE=0      never merge key splits inside the split kernel (split_counters ignored);
 *   MFA_KVCACHE_PACKED=0|1   force the kv-cache route away from / onto the packed-row kernel. */
int mfa_test_set_knob(const char* name, int value);

/* 1 while `hip_stream` is being captured into a hipGraph, 0 when not, < 0 on a HIP error (a host that owns
 * split_counters buffers must not create one during a capture). */
int mfa_stream_is_capturing(void* hip_stream);

/* Number of compute units of HIP device `device` (-1 = current); <0 on error. */
int mfa_device_cu_count(int device);

#ifdef __cplusplus
}
#endif
#endif /* MFA_H_ */
