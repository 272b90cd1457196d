// C ABI of the hot path (declared in include/mfa.h): argument checking, split-count choice and the two
// launch entry points.  No torch types; compiled into libmfa_hip.so together with the kernels.
//
// Error behaviour mirrors the reference's host layer (csrc/mfa/api.cpp): every precondition that
// api.cpp checks with TORCH_CHECK is a MFA_ERR_INVALID_ARGUMENT here, plus the ones the reference
// leaves unchecked although its kernels rely on them (16-byte alignment of pointers/strides for the
// 128-bit accesses, supported head dims, workspaces for split decode).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>

#include "mfa_launch.h"

namespace {

thread_local char g_err[512] = "";
thread_local int g_last_route = 0; // MFA_ROUTE_* bits of this thread's last successful kv-cache / forward call

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int check_common(const mfa_forward_params* p) {
    if (!p) return fail(MFA_ERR_INVALID_ARGUMENT, "params is NULL");
    if (!p->q_ptr || !p->k_ptr || !p->v_ptr || !p->o_ptr)
        return fail(MFA_ERR_INVALID_ARGUMENT, "q, k, v and o pointers must be non-NULL");
    if (p->batch < 0 || p->heads <= 0 || p->kv_heads <= 0 || p->seqlen_q < 0 || p->seqlen_k < 0)
        return fail(MFA_ERR_INVALID_ARGUMENT, "negative or zero size");
    if (p->head_dim <= 0 || p->head_dim > 256)
        return fail(MFA_ERR_INVALID_ARGUMENT, "head dimension must be less than or equal to 256");
    if (p->heads % p->kv_heads != 0)
        return fail(MFA_ERR_INVALID_ARGUMENT,
                    "number of key/value heads must be divisible by number of query heads");
    if (p->head_dim % 8 != 0)
        return fail(MFA_ERR_UNSUPPORTED, "head_dim must be a multiple of 8 (got %d)", p->head_dim);
    if (!aligned16(p->q_ptr) || !aligned16(p->k_ptr) || !aligned16(p->v_ptr) || !aligned16(p->o_ptr))
        return fail(MFA_ERR_INVALID_ARGUMENT, "q, k, v, o must be 16-byte aligned");
    const int64_t strides[] = {p->q_batch_stride, p->q_head_stride, p->q_row_stride, p->k_batch_stride,
                               p->k_head_stride,  p->k_row_stride,  p->v_batch_stride, p->v_head_stride,
                               p->v_row_stride,   p->o_batch_stride, p->o_head_stride, p->o_row_stride,
                               p->k_cache_block_stride, p->v_cache_block_stride};
    for (int64_t s : strides)
        if (s % 8 != 0)
            return fail(MFA_ERR_INVALID_ARGUMENT, "every q/k/v/o stride must be a multiple of 8 elements (16 bytes)");
    // non-paged K/V rows are addressed with 32-bit byte offsets from a per-(batch, kv head) base
    if (!p->block_table) {
        const int64_t span_k = (int64_t)p->seqlen_k * p->k_row_stride * 2, span_v = (int64_t)p->seqlen_k * p->v_row_stride * 2;
        // (varlen: seqlen_k is the longest sequence; offsets are relative to each sequence's first row)
        if (span_k >= (1LL << 32) || span_v >= (1LL << 32))
            return fail(MFA_ERR_UNSUPPORTED, "one batch element of K/V spans 4 GiB or more; use a paged cache");
    }
    if (p->block_table) {
        if (p->page_block_size <= 0) return fail(MFA_ERR_INVALID_ARGUMENT, "page_block_size must be positive");
        if (p->max_blocks_per_seq <= 0)
            return fail(MFA_ERR_INVALID_ARGUMENT, "max_blocks_per_seq must be set with a block_table");
    }
    return MFA_OK;
}

} // namespace

namespace mfa {
TestKnobs g_knobs;
}

extern "C" {

int mfa_test_set_knob(const char* name, int value) {
    if (!name) return fail(MFA_ERR_INVALID_ARGUMENT, "knob name is NULL");
    const std::pair<const char*, std::atomic<int>*> table[] = {
        {"p64_grid", &mfa::g_knobs.p64_grid},       {"group_pairs", &mfa::g_knobs.group_pairs}, {"p64_no_loop", &mfa::g_knobs.p64_no_loop},
        {"nw8", &mfa::g_knobs.nw8},                 {"mq_stream", &mfa::g_knobs.mq_stream},     {"decode_gt_max", &mfa::g_knobs.decode_gt_max}};
    for (const auto& kv : table)
        if (!std::strcmp(kv.first, name)) {
            kv.second->store(value);
            return MFA_OK;
        }
    return fail(MFA_ERR_INVALID_ARGUMENT, "unknown test knob '%s'", name);
}

int mfa_abi_version(void) { return MFA_ABI_VERSION; }

const char* mfa_version(void) { return "mini-flash-attention gfx950 0.1.0"; }

const char* mfa_last_error(void) { return g_err; }

size_t mfa_forward_params_sizeof(void) { return sizeof(mfa_forward_params); }

size_t mfa_kvcache_append_params_sizeof(void) { return sizeof(mfa_kvcache_append_params); }

int mfa_kvcache_append(const mfa_kvcache_append_params* p, void* hip_stream) {
    if (!p) return fail(MFA_ERR_INVALID_ARGUMENT, "params is NULL");
    if (!p->k_new || !p->v_new || !p->k_cache || !p->v_cache)
        return fail(MFA_ERR_INVALID_ARGUMENT, "k_new, v_new, k_cache and v_cache must be non-NULL");
    if (p->head_dim <= 0 || p->head_dim % 8 != 0 || p->head_dim > 256)
        return fail(MFA_ERR_UNSUPPORTED, "head_dim must be a multiple of 8, at most 256 (got %d)", p->head_dim);
    if (p->batch < 0 || p->seqlen_new < 0 || p->kv_heads <= 0 || p->seqlen_k < 0)
        return fail(MFA_ERR_INVALID_ARGUMENT, "negative or zero size");
    if (!aligned16(p->k_new) || !aligned16(p->v_new) || !aligned16(p->k_cache) || !aligned16(p->v_cache))
        return fail(MFA_ERR_INVALID_ARGUMENT, "k_new, v_new, k_cache, v_cache must be 16-byte aligned");
    const int64_t strides[] = {p->kn_batch_stride, p->kn_row_stride, p->kn_head_stride, p->vn_batch_stride,
                               p->vn_row_stride,   p->vn_head_stride, p->kc_batch_stride, p->kc_row_stride,
                               p->kc_head_stride,  p->vc_batch_stride, p->vc_row_stride, p->vc_head_stride};
    for (int64_t s : strides)
        if (s % 8 != 0) return fail(MFA_ERR_INVALID_ARGUMENT, "every stride must be a multiple of 8 elements (16 bytes)");
    if (p->block_table && (p->page_block_size <= 0 || p->max_blocks_per_seq <= 0))
        return fail(MFA_ERR_INVALID_ARGUMENT, "page_block_size and max_blocks_per_seq must be set with a block_table");
    if (mfa::launch_kvcache_append(*p, static_cast<hipStream_t>(hip_stream)))
        return fail(MFA_ERR_LAUNCH, "kvcache append launch failed: %s", hipGetErrorString(hipGetLastError()));
    return MFA_OK;
}

int mfa_debug_last_route(void) { return g_last_route; }

int mfa_init(int device) {
    if (device < 0 && hipGetDevice(&device) != hipSuccess) return fail(MFA_ERR_LAUNCH, "hipGetDevice failed");
    const int rc = mfa::xcd_premise_probe(device);
    if (rc < 0) return fail(MFA_ERR_LAUNCH, "mfa_init: XCD probe failed on device %d: %s", device, hipGetErrorString(hipGetLastError()));
    return rc;
}

void mfa_forward_params_set_scale(mfa_forward_params* p) {
    if (!p || p->head_dim <= 0) return;
    // reference: csrc/mfa/api.cpp:84, 99-100
    p->kv_group_size = p->kv_heads > 0 ? p->heads / p->kv_heads : 0;
    p->softmax_scale = 1.0f / std::sqrt(static_cast<float>(p->head_dim));
    p->softmax_scale_log2 = static_cast<float>(p->softmax_scale * 1.4426950408889634074);
}

int mfa_stream_is_capturing(void* hip_stream) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(static_cast<hipStream_t>(hip_stream), &cs) != hipSuccess) return fail(MFA_ERR_LAUNCH, "hipStreamIsCapturing failed");
    return cs != hipStreamCaptureStatusNone ? 1 : 0;
}

int mfa_device_cu_count(int device) {
    if (device < 0 && hipGetDevice(&device) != hipSuccess) return fail(MFA_ERR_LAUNCH, "hipGetDevice failed");
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess)
        return fail(MFA_ERR_LAUNCH, "hipDeviceGetAttribute failed");
    return n;
}

// The reference sizes its split count from batch*QUERY heads on 2*SMs (api.cpp:269-302).  The native kernel runs
// one workgroup per (batch, KV head, split) and every wave keeps 16 KiB of loads in flight, so it saturates HBM
// with well under one workgroup per CU.  Measured on MI355X (tools/split_sweep.py, profiles/r01c_sweep.txt) the time
// follows workgroup QUANTISATION over the CUs, not occupancy: W = B*Hkv*splits workgroups run at
//   eff(W) = min(1, W / (3/4 CUs))         for W <= CUs   (192 workgroups on 256 CUs already reach 6.1 TB/s)
//          = 1 - (1 - (W / CUs) / ceil(W / CUs)) / 2   for W > CUs: a CU takes a second workgroup beside the first, so a part-filled
//            last round costs about half of what whole rounds would say (384 = 1.5/CU is 10 % slower than 192 or 768; round 3,
//            tools/mha_split_sweep.py, 576 rows = 2.25/CU unsplit: Skv 512 29.9 us for 25 us of streaming, Skv 2048 108 for 101)
// and splitting costs the combine launch (~4 us) plus ~0.15 us per split of partials.  With the streaming time
// estimated as K+V bytes / 6 TB/s (head_dim 128 assumed) the count minimising  t_stream / eff + t_combine  is
// taken, never below 4 tiles of 64 keys per split, evened out.  Only the ARGUMENT semantics are the reference's:
// <1 = auto, explicit values are clamped to the number of 64-key tiles (api.cpp:320-327) and to 128.
int mfa_num_splits_heuristic(int requested, int batch, int kv_heads, int seqlen_k, int num_cus) {
    const int ntiles = (seqlen_k + 63) / 64;
    if (ntiles <= 1) return 1;
    if (requested >= 1) { // explicit: clamp to the tile count (api.cpp:325-327) and to the combine kernel's 128
        const int lim = ntiles < 128 ? ntiles : 128;
        return requested > lim ? lim : requested;
    }
    if (num_cus <= 0) {
        num_cus = mfa_device_cu_count(-1);
        if (num_cus <= 0) num_cus = 256;
    }
    const double base = static_cast<double>(batch) * kv_heads;
    if (base <= 0) return 1;
    int max_splits = ntiles / 4 > 1 ? ntiles / 4 : 1;
    if (max_splits > 128) max_splits = 128;
    int best = 1;
    double best_cost = 1e30;
    for (int s = 1; s <= max_splits; ++s) {
        const int per = (ntiles + s - 1) / s;
        if ((ntiles + per - 1) / per != s) continue; // not an even split: the evened count is evaluated on its own
        const double w = base * s;
        const double eff = w <= num_cus ? (w / (0.75 * num_cus) < 1.0 ? w / (0.75 * num_cus) : 1.0)
                                        : 1.0 - 0.5 * (1.0 - (w / num_cus) / std::ceil(w / num_cus));
        const double t_stream_us = base * seqlen_k * 512.0 / 6.0e6;
        const double cost = t_stream_us / eff + (s > 1 ? 4.0 + 0.15 * s : 0.0);
        if (cost < best_cost - 1e-9) {
            best_cost = cost;
            best = s;
        }
    }
    return best;
}

void mfa_decode_workspace_bytes(int num_splits, int batch, int heads, int head_dim, size_t* oaccum_bytes,
                                size_t* lse_bytes) {
    size_t o = 0, l = 0;
    if (num_splits > 1) {
        l = sizeof(float) * static_cast<size_t>(num_splits) * batch * heads;
        o = l * head_dim;
    }
    if (oaccum_bytes) *oaccum_bytes = o;
    if (lse_bytes) *lse_bytes = l;
}

int mfa_run_flash_attention_forward(const mfa_forward_params* p, void* hip_stream) {
    if (int rc = check_common(p)) return rc;
    if (p->head_dim % 32 != 0)
        return fail(MFA_ERR_UNSUPPORTED, "prefill supports head_dim in {32,64,...,256} (got %d)", p->head_dim);
    if (p->use_local_window && (p->local_window_left < -1 || p->local_window_right < -1))
        return fail(MFA_ERR_INVALID_ARGUMENT, "local_window_left/right must be >= -1");
    if (p->softmax_lse_ptr && p->cu_seqlens_q && p->total_q <= 0)
        return fail(MFA_ERR_INVALID_ARGUMENT, "total_q must be set to return the LSE of a varlen batch");
    if ((p->cu_seqlens_q == nullptr) != (p->cu_seqlens_k == nullptr))
        return fail(MFA_ERR_INVALID_ARGUMENT, "cu_seqlens_q and cu_seqlens_k must be given together");
    if (p->batch == 0 || p->seqlen_q == 0) return MFA_OK;
    bool p64 = false;
    const int rc = mfa::launch_prefill(*p, static_cast<hipStream_t>(hip_stream), &p64);
    if (rc == 0) g_last_route = MFA_ROUTE_PREFILL | (p64 ? MFA_ROUTE_PREFILL64 : 0);
    if (rc == -2) return fail(MFA_ERR_UNSUPPORTED, "no prefill kernel for head_dim %d", p->head_dim);
    if (rc) return fail(MFA_ERR_LAUNCH, "prefill launch failed: %s", hipGetErrorString(hipGetLastError()));
    return MFA_OK;
}

// Which kernel serves a kv-cache call (the dispatch of mfa_run_flash_attention_with_kv_cache and of mfa_kvcache_plan):
//   kKvDecode  seqlen_q == 1 and a GQA group of at most 4 (at most 2 on a paged cache): the vector kernel of
//              mfa_decode.hip (every K/V byte once, G heads in registers; 73-78 % of HBM peak);
//   kKvPacked  seqlen_q > 1, or a larger group: the G * seqlen_q query rows of a KV head packed into 32-row MFMA
//              tiles, keys split over workgroups (MQ instances of the prefill kernel);
//   kKvPrefill everything else (long query blocks, head dims without a packed instance): the prefill kernel per query
//              head over the cached keys, causal aligned to the last key.
enum { kKvDecode = 0, kKvPacked = 1, kKvPrefill = 2 };
static int kvcache_route(const mfa_forward_params* p) {
    static const int env = [] { const char* e = getenv("MFA_KVCACHE_PACKED"); return e ? atoi(e) : -1; }();
    const int g = p->kv_heads > 0 ? p->heads / p->kv_heads : 1;
    const bool has_packed = p->head_dim == 32 || p->head_dim == 64 || p->head_dim == 96 || p->head_dim == 128 || p->head_dim == 256;
    const int64_t rows = static_cast<int64_t>(p->seqlen_q) * g;
    // (paged caches, groups of 3-4: the packed kernel's tile DMA holds 5.3-5.6 TB/s where the vector kernel does 4.8
    //  (profiles/r01d_kvcache_paged_routes.txt, r02 re-measured with the vector kernel's page-aligned mode: 4.76 vs
    //  5.29 on BASELINE config 5) -- but only when a 64-key tile lies in one page; with smaller or odd page sizes the
    //  packed kernel looks a page up per staged row and falls to 3.0 TB/s, the vector kernel keeps 5.1)
    const int ps = p->page_block_size;
    const bool paged_group = p->block_table != nullptr && g >= 3 && ps >= 64 && (ps & (ps - 1)) == 0;
    bool packed = has_packed && rows <= 512 && (p->seqlen_q > 1 || g > 4 || p->use_local_window || paged_group);
    // A long query block on few heads over a long cache (a prompt chunk on a tensor-parallel shard's one KV head): the per-head
    // prefill kernel has batch * heads * ceil(Sq / 128) workgroups and no way to split the keys; with at most one for every
    // second CU the packed kernel takes it, whose key splits fill the chip (bf16 B1 Hq8 Hkv1 Sq2048 Skv32768: 541 -> 283 us;
    // with more workgroups than that the per-head kernel is the faster one, tools/chunked_prefill_point.py)
    if (!packed && has_packed && p->seqlen_q > 1 && p->seqlen_k >= 4096) {
        const int cus = p->num_cus > 0 ? p->num_cus : 256;
        const int64_t wgs = static_cast<int64_t>(p->batch) * p->heads * ((p->seqlen_q + 127) / 128);
        const int64_t span = static_cast<int64_t>(p->seqlen_q) * std::max(p->q_row_stride, p->o_row_stride) +
                             static_cast<int64_t>(g) * std::max(p->q_head_stride, p->o_head_stride);
        packed = 2 * wgs <= cus && span < (1LL << 31);
    }
    if (env == 0) packed = false;
    if (env == 1 && has_packed) packed = true;
    if (packed) return kKvPacked;
    return p->seqlen_q == 1 && !p->use_local_window ? kKvDecode : kKvPrefill; // (the vector kernel has no window)
}

// key splits for the packed kernel: W = B*Hkv*row_blocks*splits workgroups run in ceil(W / slots) rounds of
// (tiles per split + ~3 tiles of fixed cost); splitting adds the combine launch
static int packed_num_splits(const mfa_forward_params* p, int num_cus) {
    const int ntiles = (p->seqlen_k + 63) / 64;
    const int g = p->heads / p->kv_heads;
    const int64_t items = static_cast<int64_t>(p->batch) * p->kv_heads * ((static_cast<int64_t>(p->seqlen_q) * g + 127) / 128);
    if (num_cus <= 0) num_cus = mfa_device_cu_count(-1);
    if (num_cus <= 0) num_cus = 256;
    const int64_t slots = static_cast<int64_t>(num_cus) * (p->head_dim <= 128 ? 2 : 1);
    int best = 1;
    double best_cost = 1e300;
    const int smax = std::max(1, std::min(128, ntiles / 4));
    for (int s = 1; s <= smax; ++s) {
        const double rounds = std::ceil(static_cast<double>(items * s) / slots);
        const double cost = rounds * ((ntiles + s - 1) / s + 3.0) + (s > 1 ? 3.0 + 0.1 * s : 0.0);
        if (cost < best_cost - 1e-9) {
            best_cost = cost;
            best = s;
        }
    }
    return best;
}

int mfa_kvcache_plan(const mfa_forward_params* p, int* num_splits, size_t* oaccum_bytes, size_t* lse_bytes) {
    if (!p || p->kv_heads <= 0 || p->heads % p->kv_heads != 0 || p->seqlen_q < 1)
        return fail(MFA_ERR_INVALID_ARGUMENT, "mfa_kvcache_plan: heads / kv_heads / seqlen_q not set");
    const int ntiles = std::max(1, (p->seqlen_k + 63) / 64);
    int s = 1;
    switch (kvcache_route(p)) {
    case kKvDecode: s = mfa_num_splits_heuristic(p->num_splits, p->batch, p->kv_heads, p->seqlen_k, p->num_cus); break;
    case kKvPacked: s = p->num_splits >= 1 ? std::min(std::min(p->num_splits, ntiles), 128) : packed_num_splits(p, p->num_cus); break;
    default: s = 1; break;
    }
    if (num_splits) *num_splits = s;
    size_t o = 0, l = 0;
    if (s > 1) {
        l = sizeof(float) * static_cast<size_t>(s) * p->batch * p->seqlen_q * p->heads;
        o = l * p->head_dim;
    }
    if (oaccum_bytes) *oaccum_bytes = o;
    if (lse_bytes) *lse_bytes = l;
    return MFA_OK;
}

// counters of the in-kernel merge for this problem (the launchers use the same arithmetic): one per (batch, KV head, head
// chunk) for the vector kernel, one per (batch, KV head, block of 128 packed rows) for the packed kernel
static int64_t decode_head_chunks(int g) { // (mfa_decode.hip launch_decode: group tiles of 1-4, 6, 8 heads)
    const int gt = g <= 4 ? g : (g <= 6 ? 6 : 8);
    return (g + gt - 1) / gt;
}
size_t mfa_kvcache_counter_count(const mfa_forward_params* p) {
    if (!p || p->num_splits <= 1 || p->kv_heads <= 0 || p->heads % p->kv_heads != 0 || p->seqlen_q < 1) return 0;
    const int g = p->heads / p->kv_heads;
    int64_t n = 0, units = 0;
    mfa_forward_params one = *p;
    switch (kvcache_route(p)) {
    case kKvDecode: units = n = (int64_t)p->batch * p->kv_heads * decode_head_chunks(g); one.seqlen_q = 1; break;
    case kKvPacked: units = (int64_t)p->batch * p->kv_heads; n = units * (((int64_t)p->seqlen_q * g + 127) / 128); break;
    default: return 0;
    }
    if (n > MFA_SPLIT_COUNTERS_MAX || !mfa::fused_merge_pays(units, n * p->num_splits, mfa::partial_bytes(one))) return 0;
    return (size_t)n;
}

int mfa_run_flash_attention_with_kv_cache(const mfa_forward_params* p, void* hip_stream) {
    if (int rc = check_common(p)) return rc;
    if (p->seqlen_q < 1) return fail(MFA_ERR_INVALID_ARGUMENT, "seqlen_q must be >= 1, got %d", p->seqlen_q);
    if (p->cu_seqlens_q || p->cu_seqlens_k)
        return fail(MFA_ERR_INVALID_ARGUMENT, "the kv-cache entry takes (B, Sq, H, D) queries, not packed sequences");
    if (p->num_splits > 128)
        return fail(MFA_ERR_INVALID_ARGUMENT, "num_splits must be <= 128 (got %d); see mfa_kvcache_plan", p->num_splits);
    if (p->num_splits > 1 && (!p->softmax_lseaccum_ptr || !p->oaccum_ptr))
        return fail(MFA_ERR_WORKSPACE, "num_splits=%d needs softmax_lseaccum_ptr and oaccum_ptr", p->num_splits);
    if (p->use_local_window && (p->local_window_left < -1 || p->local_window_right < -1))
        return fail(MFA_ERR_INVALID_ARGUMENT, "local_window_left/right must be >= -1");
    // O may be strided: the combine kernel honours o_*_stride (the reference assumes contiguous, decode.cuh:730)
    if (p->batch == 0) return MFA_OK;
    const hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    const int route = kvcache_route(p);
    if (route == kKvDecode) {
        bool fused = false;
        const int rc = mfa::launch_decode(*p, stream, &fused);
        if (rc == -4)
            return fail(MFA_ERR_UNSUPPORTED, "decode launches batch x kv_heads x head chunks x splits workgroups: the product must stay "
                        "below 2^30 (batch %d)", p->batch);
        if (rc) return fail(MFA_ERR_LAUNCH, "decode launch failed: %s", hipGetErrorString(hipGetLastError()));
        g_last_route = MFA_ROUTE_DECODE | (p->num_splits > 1 ? (fused ? MFA_ROUTE_FUSED_MERGE : MFA_ROUTE_COMBINE_LAUNCH) : 0);
        return MFA_OK;
    }
    // the queries are the LAST seqlen_q positions of the sequence: causal / windows align to the last key
    mfa_forward_params q = *p;
    q.mask_bottom_right = 1;
    if (p->seqlen_q == 1 && !p->use_local_window) q.is_causal = 0; // one query at the end sees every key
    if (route == kKvPrefill) {
        if (p->head_dim % 32 != 0)
            return fail(MFA_ERR_UNSUPPORTED, "kv-cache attention with seqlen_q > 1 needs head_dim %% 32 == 0 (got %d)", p->head_dim);
        if (p->num_splits > 1) return fail(MFA_ERR_INVALID_ARGUMENT, "this shape runs unsplit: ask mfa_kvcache_plan for num_splits");
        bool p64 = false;
        const int rc = mfa::launch_prefill(q, stream, &p64);
        if (rc) return fail(MFA_ERR_LAUNCH, "kv-cache prefill launch failed: %s", hipGetErrorString(hipGetLastError()));
        g_last_route = MFA_ROUTE_PREFILL | (p64 ? MFA_ROUTE_PREFILL64 : 0);
        return MFA_OK;
    }
    bool fused = false;
    const int rc = mfa::launch_kvcache_packed(q, stream, &fused);
    if (rc) return fail(MFA_ERR_LAUNCH, "packed kv-cache launch failed: %s", hipGetErrorString(hipGetLastError()));
    g_last_route = MFA_ROUTE_PACKED | (p->num_splits > 1 ? (fused ? MFA_ROUTE_FUSED_MERGE : MFA_ROUTE_COMBINE_LAUNCH) : 0);
    return MFA_OK;
}

} // extern "C"
