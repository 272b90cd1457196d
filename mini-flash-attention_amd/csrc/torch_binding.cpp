// mini_flash_attention._C — the PyTorch-ROCm extension module with the reference's ABI:
//   mini_flash_attention_forward / _varlen_forward / _with_kvcache     (reference csrc/api.cpp:4-9)
// Host op layer = the reference's csrc/mfa/api.cpp re-done above the C ABI of include/mfa.h: validate,
// allocate the output (and split workspaces) from the caching allocator, fill mfa_forward_params, launch
// on the current HIP stream.  This file is the ONLY torch-aware native code; it never computes.
#include <ATen/ATen.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/extension.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <optional>
#include <tuple>
#include <utility>

#include "mfa.h"

namespace {

#define MFA_CHECK_DEVICE(x) TORCH_CHECK(x.is_cuda(), #x " must be on CUDA")
// every tensor argument must live on q's device: the kernels run on q's device and stream with raw pointers
#define MFA_CHECK_SAME_DEVICE(x, ref) \
    TORCH_CHECK(x.is_cuda() && x.device() == ref.device(), #x " must be on the same device as " #ref)
#define MFA_CHECK_SHAPE(x, ...) \
    TORCH_CHECK(x.sizes() == at::IntArrayRef({__VA_ARGS__}), #x " must have shape (" #__VA_ARGS__ ")")

void* current_stream(const at::Tensor& t) {
    return static_cast<void*>(c10::hip::getCurrentHIPStream(t.device().index()).stream());
}

void check_rc(int rc) { TORCH_CHECK(rc == MFA_OK, "mini_flash_attention: ", mfa_last_error(), " (code ", rc, ")"); }

void check_dtypes(const at::Tensor& q, const at::Tensor& k, const at::Tensor& v) {
    auto dtype = q.scalar_type();
    TORCH_CHECK(dtype == at::kHalf || dtype == at::kBFloat16, "FlashAttention only support fp16 and bf16 data type");
    TORCH_CHECK(k.scalar_type() == dtype, "query and key must have the same dtype");
    TORCH_CHECK(v.scalar_type() == dtype, "query and value must have the same dtype");
    MFA_CHECK_DEVICE(q);
    MFA_CHECK_SAME_DEVICE(k, q);
    MFA_CHECK_SAME_DEVICE(v, q);
    TORCH_CHECK(q.stride(-1) == 1, "Input tensor must have contiguous last dimension");
    TORCH_CHECK(k.stride(-1) == 1, "Input tensor must have contiguous last dimension");
    TORCH_CHECK(v.stride(-1) == 1, "Input tensor must have contiguous last dimension");
}

// strides as the reference takes them (api.cpp:58-74): row = stride(-3), head = stride(-2), batch = stride(0)
void set_tensor_strides(mfa_forward_params& p, const at::Tensor& q, const at::Tensor& k, const at::Tensor& v,
                        const at::Tensor& o, bool has_batch_dim) {
    p.q_ptr = q.data_ptr();
    p.k_ptr = k.data_ptr();
    p.v_ptr = v.data_ptr();
    p.o_ptr = o.data_ptr();
    p.q_row_stride = q.stride(-3);
    p.k_row_stride = k.stride(-3);
    p.v_row_stride = v.stride(-3);
    p.o_row_stride = o.stride(-3);
    p.q_head_stride = q.stride(-2);
    p.k_head_stride = k.stride(-2);
    p.v_head_stride = v.stride(-2);
    p.o_head_stride = o.stride(-2);
    if (has_batch_dim) {
        p.q_batch_stride = q.stride(0);
        p.k_batch_stride = k.stride(0);
        p.v_batch_stride = v.stride(0);
        p.o_batch_stride = o.stride(0);
    }
    p.is_bf16 = q.scalar_type() == at::kBFloat16;
}

void set_windows(mfa_forward_params& p, int wl, int wr, int seqlen_k) {
    // api.cpp:88-96: accepted and normalised, only is_causal has an effect
    p.is_causal = wl < 0 && wr == 0;
    if (wl < 0 && wr >= 0) wl = seqlen_k;
    if (wl >= 0 && wr < 0) wr = seqlen_k;
    p.window_size_left = wl;
    p.window_size_right = wr;
}

void set_paged(mfa_forward_params& p, const at::Tensor& block_table, const at::Tensor& k, const at::Tensor& v,
               int batch) {
    MFA_CHECK_SAME_DEVICE(block_table, k);
    TORCH_CHECK(block_table.scalar_type() == at::kInt, "block_table must be int32");
    TORCH_CHECK(block_table.dim() == 2 && block_table.size(0) == batch,
                "block_table must have the same batch size as q");
    TORCH_CHECK(block_table.stride(-1) == 1, "block_table must have contiguous last dimension");
    p.block_table = block_table.data_ptr<int>();
    p.block_table_batch_stride = block_table.stride(0);
    p.max_blocks_per_seq = block_table.size(1);
    p.page_block_size = k.size(1);
    p.k_cache_block_stride = k.stride(0);
    p.v_cache_block_stride = v.stride(0);
}

// kv-cache launch: asks the library for its key-split count and workspace sizes (reference:
// forward_params_set_split_kv, api.cpp:305-340; the kernels write -inf LSE for empty splits themselves, so no fill
// kernel is launched), allocates them from the caching allocator and runs.
//
// Arrival counters of the in-kernel split merge (mfa_forward_params::split_counters): the C ABI never allocates, so this
// layer owns them -- one zeroed MFA_SPLIT_COUNTERS_MAX-entry int32 tensor per (device, stream), created at the first split
// call on that stream together with the once-per-device mfa_init() probe, kept for the life of the process (never grown or
// freed: a hipGraph that captured a launch keeps a valid pointer).  While the stream is being captured nothing is created:
// a launch without counters merges through decode_combine_kernel, which captures like any other launch.
static int32_t* split_counters_for(const at::Tensor& q, void* stream) {
    static std::mutex mu;
    static std::map<std::pair<int, void*>, at::Tensor> bufs;
    static std::map<int, int> inited;
    const int dev = q.device().index();
    std::lock_guard<std::mutex> lock(mu);
    auto it = bufs.find({dev, stream});
    if (it != bufs.end()) return it->second.data_ptr<int32_t>();
    if (mfa_stream_is_capturing(stream) != 0) return nullptr;
    if (!inited.count(dev)) inited[dev] = mfa_init(dev);
    if (inited[dev] != 1) return nullptr;
    at::Tensor t = at::zeros({MFA_SPLIT_COUNTERS_MAX}, q.options().dtype(at::kInt));
    bufs.emplace(std::make_pair(dev, stream), t);
    return t.data_ptr<int32_t>();
}

static void run_kvcache(mfa_forward_params& p, const at::Tensor& q, int requested_splits) {
    p.num_splits = requested_splits;
    int splits = 1;
    size_t o_bytes = 0, lse_bytes = 0;
    check_rc(mfa_kvcache_plan(&p, &splits, &o_bytes, &lse_bytes));
    p.num_splits = splits;
    at::Tensor lse_accum, out_accum;
    void* stream = current_stream(q);
    if (splits > 1) {
        auto opts = q.options().dtype(at::kFloat);
        lse_accum = at::empty({static_cast<int64_t>(lse_bytes / sizeof(float))}, opts);
        out_accum = at::empty({static_cast<int64_t>(o_bytes / sizeof(float))}, opts);
        p.softmax_lseaccum_ptr = lse_accum.data_ptr<float>();
        p.oaccum_ptr = out_accum.data_ptr<float>();
        if (mfa_kvcache_counter_count(&p) > 0) {
            p.split_counters = split_counters_for(q, stream);
            p.split_counters_len = p.split_counters ? MFA_SPLIT_COUNTERS_MAX : 0;
        }
    }
    check_rc(mfa_run_flash_attention_with_kv_cache(&p, stream));
}

// reference: mfa::flash_attention_forward, csrc/mfa/api.cpp:113-186
at::Tensor flash_attention_forward(const at::Tensor& q, const at::Tensor& k, const at::Tensor& v,
                                   std::optional<at::Tensor> out_, bool is_causal, int window_size_left,
                                   int window_size_right) {
    check_dtypes(q, k, v);
    c10::DeviceGuard guard(q.device());
    TORCH_CHECK(q.dim() == 4 && k.dim() == 4 && v.dim() == 4, "q, k, v must be 4-D (batch, seqlen, heads, head_dim)");
    const int batch = q.size(0), seqlen_q = q.size(1), num_heads = q.size(2), head_dim = q.size(3);
    const int seqlen_k = k.size(1), kv_num_heads = k.size(2);
    TORCH_CHECK(k.size(0) == batch, "batch size of q and k must be the same");
    TORCH_CHECK(v.size(0) == batch, "batch size of q and v must be the same");
    TORCH_CHECK(v.size(1) == seqlen_k, "sequence length of k must be the same as v");
    TORCH_CHECK(head_dim <= 256, "head dimension must be less than or equal to 256");
    TORCH_CHECK(kv_num_heads > 0 && num_heads % kv_num_heads == 0,
                "number of key/value heads must be divisible by number of query heads");
    if (window_size_left >= seqlen_k) window_size_left = -1;
    if (window_size_right >= seqlen_k) window_size_right = -1;
    if (is_causal) window_size_right = 0;
    MFA_CHECK_SHAPE(q, batch, seqlen_q, num_heads, head_dim);
    MFA_CHECK_SHAPE(k, batch, seqlen_k, kv_num_heads, head_dim);
    MFA_CHECK_SHAPE(v, batch, seqlen_k, kv_num_heads, head_dim);

    at::Tensor out;
    if (out_.has_value()) {
        out = out_.value();
        TORCH_CHECK(out.scalar_type() == q.scalar_type(), "Output tensor must have the dtype of q");
        MFA_CHECK_SAME_DEVICE(out, q);
        TORCH_CHECK(out.stride(-1) == 1, "Output tensor must have contiguous last dimension");
        MFA_CHECK_SHAPE(out, batch, seqlen_q, num_heads, head_dim);
    } else {
        out = at::empty_like(q);
    }

    mfa_forward_params p{};
    set_tensor_strides(p, q, k, v, out, true);
    p.batch = batch;
    p.seqlen_q = seqlen_q;
    p.seqlen_k = seqlen_k;
    p.heads = num_heads;
    p.kv_heads = kv_num_heads;
    p.head_dim = head_dim;
    mfa_forward_params_set_scale(&p);
    set_windows(p, window_size_left, window_size_right, seqlen_k);
    check_rc(mfa_run_flash_attention_forward(&p, current_stream(q)));
    return out;
}

// reference: mfa::flash_attention_varlen_forward, csrc/mfa/api.cpp:189-267
at::Tensor flash_attention_varlen_forward(const at::Tensor& q, const at::Tensor& k, const at::Tensor& v,
                                          const at::Tensor& cu_seqlens_q, const at::Tensor& cu_seqlens_k,
                                          const int max_seqlen_q, const int max_seqlen_k, bool is_causal,
                                          int window_size_left, int window_size_right,
                                          const std::optional<at::Tensor>& block_table_) {
    check_dtypes(q, k, v);
    c10::DeviceGuard guard(q.device());
    MFA_CHECK_SAME_DEVICE(cu_seqlens_q, q);
    MFA_CHECK_SAME_DEVICE(cu_seqlens_k, q);
    TORCH_CHECK(cu_seqlens_q.scalar_type() == at::kInt && cu_seqlens_k.scalar_type() == at::kInt,
                "cu_seqlens_q and cu_seqlens_k must be int32");
    TORCH_CHECK(cu_seqlens_q.is_contiguous() && cu_seqlens_k.is_contiguous(), "cu_seqlens must be contiguous");
    TORCH_CHECK(cu_seqlens_q.numel() == cu_seqlens_k.numel() && cu_seqlens_q.numel() >= 1,
                "cu_seqlens_q and cu_seqlens_k must have batch + 1 elements");
    TORCH_CHECK(q.dim() == 3, "q must be (total_q, heads, head_dim)");
    const int total_q = q.size(0), num_heads = q.size(1), head_dim = q.size(2);
    const int batch = cu_seqlens_q.numel() - 1;
    const int kv_num_heads = k.size(-2);
    TORCH_CHECK(head_dim <= 256, "head dimension must be less than or equal to 256");
    TORCH_CHECK(kv_num_heads > 0 && num_heads % kv_num_heads == 0,
                "number of key/value heads must be divisible by number of query heads");
    at::Tensor out = at::empty_like(q);
    if (is_causal) window_size_right = 0;

    mfa_forward_params p{};
    set_tensor_strides(p, q, k, v, out, false);
    p.batch = batch;
    p.seqlen_q = max_seqlen_q;
    p.seqlen_k = max_seqlen_k;
    p.heads = num_heads;
    p.kv_heads = kv_num_heads;
    p.head_dim = head_dim;
    p.cu_seqlens_q = cu_seqlens_q.data_ptr<int>();
    p.cu_seqlens_k = cu_seqlens_k.data_ptr<int>();
    mfa_forward_params_set_scale(&p);
    set_windows(p, window_size_left, window_size_right, max_seqlen_k);

    MFA_CHECK_SHAPE(q, total_q, num_heads, head_dim);
    p.total_q = total_q; // (the launcher reads how even the batch is off total_q / batch against max_seqlen_q)
    if (block_table_.has_value()) {
        TORCH_CHECK(k.dim() == 4 && v.dim() == 4, "paged k, v must be (num_blocks, page_block_size, heads_k, head_dim)");
        const int num_blocks = k.size(0), page_block_size = k.size(1);
        MFA_CHECK_SHAPE(k, num_blocks, page_block_size, kv_num_heads, head_dim);
        MFA_CHECK_SHAPE(v, num_blocks, page_block_size, kv_num_heads, head_dim);
        set_paged(p, block_table_.value(), k, v, batch);
    } else {
        // the reference requires total_k == total_q here (api.cpp:259-260); any total_k is accepted
        TORCH_CHECK(k.dim() == 3 && v.dim() == 3, "k, v must be (total_k, heads_k, head_dim)");
        const int total_k = k.size(0);
        MFA_CHECK_SHAPE(k, total_k, kv_num_heads, head_dim);
        MFA_CHECK_SHAPE(v, total_k, kv_num_heads, head_dim);
        // K/V rows are addressed with 32-bit byte offsets from each sequence's first row; the library checks that with
        // max_seqlen_k, which the caller could under-report: bound the span with the tensor itself.  (max_seqlen_q /
        // max_seqlen_k must be >= the longest sequence in cu_seqlens: rows past an under-reported maximum are not computed.)
        TORCH_CHECK(static_cast<int64_t>(total_k) * std::max(k.stride(0), v.stride(0)) * 2 < (1LL << 32),
                    "varlen k/v span 4 GiB or more: split the batch or use a paged cache");
    }
    check_rc(mfa_run_flash_attention_forward(&p, current_stream(q)));
    return out;
}

// reference: mfa::mha_fwd_kvcache, csrc/mfa/api.cpp:343-446
at::Tensor mha_fwd_kvcache(const at::Tensor& q, const at::Tensor& k_cache, const at::Tensor& v_cache,
                           const std::optional<at::Tensor>& seqlens_k_,
                           const std::optional<at::Tensor>& block_table_, bool /*causal: ignored, api.cpp:349*/,
                           int num_splits) {
    check_dtypes(q, k_cache, v_cache);
    c10::DeviceGuard guard(q.device());
    TORCH_CHECK(q.dim() == 4 && k_cache.dim() == 4 && v_cache.dim() == 4, "q, k_cache, v_cache must be 4-D");
    const int batch = q.size(0), seqlen_q = q.size(1), num_heads = q.size(2), head_dim = q.size(3);
    TORCH_CHECK(seqlen_q == 1, "flash decoding expects seqlen_q == 1, got ", seqlen_q);
    const bool paged_kv = block_table_.has_value();
    const int kv_num_heads = k_cache.size(-2);
    TORCH_CHECK(head_dim <= 256, "head dimension must be less than or equal to 256");
    TORCH_CHECK(kv_num_heads > 0 && num_heads % kv_num_heads == 0,
                "number of key/value heads must be divisible by number of query heads");
    MFA_CHECK_SHAPE(q, batch, seqlen_q, num_heads, head_dim);

    mfa_forward_params p{};
    int seqlen_k;
    if (paged_kv) {
        const auto& block_table = block_table_.value();
        TORCH_CHECK(block_table.dim() == 2, "block_table must be (batch, max_blocks_per_seq)");
        const int page_block_size = k_cache.size(1), num_blocks = k_cache.size(0);
        seqlen_k = block_table.size(1) * page_block_size;
        MFA_CHECK_SHAPE(k_cache, num_blocks, page_block_size, kv_num_heads, head_dim);
        MFA_CHECK_SHAPE(v_cache, num_blocks, page_block_size, kv_num_heads, head_dim);
        set_paged(p, block_table, k_cache, v_cache, batch);
    } else {
        seqlen_k = k_cache.size(1);
        MFA_CHECK_SHAPE(k_cache, batch, seqlen_k, kv_num_heads, head_dim);
        MFA_CHECK_SHAPE(v_cache, batch, seqlen_k, kv_num_heads, head_dim);
    }
    at::Tensor out = at::empty_like(q);
    set_tensor_strides(p, q, k_cache, v_cache, out, true);
    p.batch = batch;
    p.seqlen_q = seqlen_q;
    p.seqlen_k = seqlen_k;
    p.heads = num_heads;
    p.kv_heads = kv_num_heads;
    p.head_dim = head_dim;
    mfa_forward_params_set_scale(&p);
    set_windows(p, -1, -1, seqlen_k);

    if (seqlens_k_.has_value()) {
        const auto& seqlens_k = seqlens_k_.value();
        MFA_CHECK_SAME_DEVICE(seqlens_k, q);
        TORCH_CHECK(seqlens_k.scalar_type() == at::kInt, "seqlens_k must be int32");
        TORCH_CHECK(seqlens_k.numel() == batch, "seqlens_k must have the same number of elements as batch size");
        TORCH_CHECK(seqlens_k.is_contiguous(), "seqlens_k must be contiguous");
        p.seqlens_k = seqlens_k.data_ptr<int>();
    } // None = every sequence is seqlen_k long (the reference dereferences NULL here, decode.cuh:26)

    auto opts = q.options().dtype(at::kFloat);
    at::Tensor softmax_lse = at::empty({batch, num_heads}, opts);
    p.softmax_lse_ptr = softmax_lse.data_ptr<float>();

    run_kvcache(p, q, num_splits);
    return out;
}


// ---------------------------------------------------------------------------------------------------------------
// Extended entry points: the SURVEY.md §8(f) "next" rows, opt-in supersets of the reference's three functions.
//   * sliding-window (local) attention: keys in [row + off - left, row + off + right]
//   * the natural-log LSE as a second output
//   * kv-cache attention with new-token append (k=, v=) and seqlen_q > 1 (bottom-right aligned causal)
// The reference-ABI functions above are untouched (they accept and ignore window sizes, as upstream).
// ---------------------------------------------------------------------------------------------------------------
using OutLse = std::tuple<at::Tensor, std::optional<at::Tensor>>;

void set_extras(mfa_forward_params& p, int window_left, int window_right, bool bottom_right) {
    TORCH_CHECK(window_left >= -1 && window_right >= -1, "window sizes must be >= -1 (-1 = unbounded)");
    if (window_left >= 0 || window_right >= 0) {
        p.use_local_window = 1;
        p.local_window_left = window_left;
        p.local_window_right = window_right;
    }
    p.mask_bottom_right = bottom_right;
}

OutLse forward_ex(const at::Tensor& q, const at::Tensor& k, const at::Tensor& v, std::optional<at::Tensor> out_,
                  bool is_causal, int window_left, int window_right, bool bottom_right, bool return_lse) {
    check_dtypes(q, k, v);
    c10::DeviceGuard guard(q.device());
    TORCH_CHECK(q.dim() == 4 && k.dim() == 4 && v.dim() == 4, "q, k, v must be 4-D (batch, seqlen, heads, head_dim)");
    const int batch = q.size(0), seqlen_q = q.size(1), num_heads = q.size(2), head_dim = q.size(3);
    const int seqlen_k = k.size(1), kv_num_heads = k.size(2);
    TORCH_CHECK(head_dim <= 256, "head dimension must be less than or equal to 256");
    TORCH_CHECK(kv_num_heads > 0 && num_heads % kv_num_heads == 0,
                "number of key/value heads must be divisible by number of query heads");
    MFA_CHECK_SHAPE(q, batch, seqlen_q, num_heads, head_dim);
    MFA_CHECK_SHAPE(k, batch, seqlen_k, kv_num_heads, head_dim);
    MFA_CHECK_SHAPE(v, batch, seqlen_k, kv_num_heads, head_dim);
    at::Tensor out;
    if (out_.has_value()) {
        out = out_.value();
        TORCH_CHECK(out.scalar_type() == q.scalar_type() && out.is_cuda() && out.device() == q.device() && out.stride(-1) == 1, "bad out tensor");
        MFA_CHECK_SHAPE(out, batch, seqlen_q, num_heads, head_dim);
    } else {
        out = at::empty_like(q);
    }
    mfa_forward_params p{};
    set_tensor_strides(p, q, k, v, out, true);
    p.batch = batch; p.seqlen_q = seqlen_q; p.seqlen_k = seqlen_k;
    p.heads = num_heads; p.kv_heads = kv_num_heads; p.head_dim = head_dim;
    mfa_forward_params_set_scale(&p);
    set_windows(p, -1, is_causal ? 0 : -1, seqlen_k);
    set_extras(p, window_left, window_right, bottom_right);
    std::optional<at::Tensor> lse;
    if (return_lse) {
        lse = at::empty({batch, num_heads, seqlen_q}, q.options().dtype(at::kFloat));
        p.softmax_lse_ptr = lse->data_ptr<float>();
    }
    check_rc(mfa_run_flash_attention_forward(&p, current_stream(q)));
    return {out, lse};
}

OutLse varlen_forward_ex(const at::Tensor& q, const at::Tensor& k, const at::Tensor& v, const at::Tensor& cu_seqlens_q,
                         const at::Tensor& cu_seqlens_k, int max_seqlen_q, int max_seqlen_k, bool is_causal,
                         int window_left, int window_right, bool bottom_right, bool return_lse,
                         const std::optional<at::Tensor>& block_table_) {
    check_dtypes(q, k, v);
    c10::DeviceGuard guard(q.device());
    MFA_CHECK_SAME_DEVICE(cu_seqlens_q, q);
    MFA_CHECK_SAME_DEVICE(cu_seqlens_k, q);
    TORCH_CHECK(cu_seqlens_q.scalar_type() == at::kInt && cu_seqlens_k.scalar_type() == at::kInt,
                "cu_seqlens_q and cu_seqlens_k must be int32");
    TORCH_CHECK(cu_seqlens_q.is_contiguous() && cu_seqlens_k.is_contiguous(), "cu_seqlens must be contiguous");
    TORCH_CHECK(cu_seqlens_q.numel() == cu_seqlens_k.numel() && cu_seqlens_q.numel() >= 1,
                "cu_seqlens_q and cu_seqlens_k must have batch + 1 elements");
    TORCH_CHECK(q.dim() == 3, "q must be (total_q, heads, head_dim)");
    const int total_q = q.size(0), num_heads = q.size(1), head_dim = q.size(2);
    const int batch = cu_seqlens_q.numel() - 1, kv_num_heads = k.size(-2);
    TORCH_CHECK(head_dim <= 256, "head dimension must be less than or equal to 256");
    TORCH_CHECK(kv_num_heads > 0 && num_heads % kv_num_heads == 0,
                "number of key/value heads must be divisible by number of query heads");
    at::Tensor out = at::empty_like(q);
    mfa_forward_params p{};
    set_tensor_strides(p, q, k, v, out, false);
    p.batch = batch; p.seqlen_q = max_seqlen_q; p.seqlen_k = max_seqlen_k;
    p.heads = num_heads; p.kv_heads = kv_num_heads; p.head_dim = head_dim;
    p.cu_seqlens_q = cu_seqlens_q.data_ptr<int>();
    p.cu_seqlens_k = cu_seqlens_k.data_ptr<int>();
    p.total_q = total_q;
    mfa_forward_params_set_scale(&p);
    set_windows(p, -1, is_causal ? 0 : -1, max_seqlen_k);
    set_extras(p, window_left, window_right, bottom_right);
    if (block_table_.has_value()) {
        TORCH_CHECK(k.dim() == 4 && v.dim() == 4, "paged k, v must be (num_blocks, page_block_size, heads_k, head_dim)");
        MFA_CHECK_SHAPE(k, k.size(0), k.size(1), kv_num_heads, head_dim);
        MFA_CHECK_SHAPE(v, k.size(0), k.size(1), kv_num_heads, head_dim);
        set_paged(p, block_table_.value(), k, v, batch);
    } else {
        TORCH_CHECK(k.dim() == 3 && v.dim() == 3, "k, v must be (total_k, heads_k, head_dim)");
        MFA_CHECK_SHAPE(k, k.size(0), kv_num_heads, head_dim);
        MFA_CHECK_SHAPE(v, k.size(0), kv_num_heads, head_dim);
        TORCH_CHECK(static_cast<int64_t>(k.size(0)) * std::max(k.stride(0), v.stride(0)) * 2 < (1LL << 32),
                    "varlen k/v span 4 GiB or more: split the batch or use a paged cache");
    }
    std::optional<at::Tensor> lse;
    if (return_lse) {
        lse = at::empty({num_heads, total_q}, q.options().dtype(at::kFloat));
        p.softmax_lse_ptr = lse->data_ptr<float>();
    }
    check_rc(mfa_run_flash_attention_forward(&p, current_stream(q)));
    return {out, lse};
}

OutLse kvcache_ex(const at::Tensor& q, const at::Tensor& k_cache, const at::Tensor& v_cache,
                  const std::optional<at::Tensor>& k_new_, const std::optional<at::Tensor>& v_new_,
                  const std::optional<at::Tensor>& seqlens_k_, const std::optional<at::Tensor>& block_table_, bool causal,
                  int window_left, int window_right, int num_splits, bool return_lse) {
    check_dtypes(q, k_cache, v_cache);
    c10::DeviceGuard guard(q.device());
    TORCH_CHECK(q.dim() == 4 && k_cache.dim() == 4 && v_cache.dim() == 4, "q, k_cache, v_cache must be 4-D");
    const int batch = q.size(0), seqlen_q = q.size(1), num_heads = q.size(2), head_dim = q.size(3);
    const bool paged_kv = block_table_.has_value();
    const int kv_num_heads = k_cache.size(-2);
    TORCH_CHECK(seqlen_q >= 1, "seqlen_q must be at least 1");
    TORCH_CHECK(head_dim <= 256, "head dimension must be less than or equal to 256");
    TORCH_CHECK(kv_num_heads > 0 && num_heads % kv_num_heads == 0,
                "number of key/value heads must be divisible by number of query heads");
    MFA_CHECK_SHAPE(q, batch, seqlen_q, num_heads, head_dim);
    mfa_forward_params p{};
    int seqlen_k;
    if (paged_kv) {
        const auto& block_table = block_table_.value();
        TORCH_CHECK(block_table.dim() == 2, "block_table must be (batch, max_blocks_per_seq)");
        seqlen_k = block_table.size(1) * k_cache.size(1);
        MFA_CHECK_SHAPE(k_cache, k_cache.size(0), k_cache.size(1), kv_num_heads, head_dim);
        MFA_CHECK_SHAPE(v_cache, k_cache.size(0), k_cache.size(1), kv_num_heads, head_dim);
        set_paged(p, block_table, k_cache, v_cache, batch);
    } else {
        seqlen_k = k_cache.size(1);
        MFA_CHECK_SHAPE(k_cache, batch, seqlen_k, kv_num_heads, head_dim);
        MFA_CHECK_SHAPE(v_cache, batch, seqlen_k, kv_num_heads, head_dim);
    }
    const int32_t* seqlens_ptr = nullptr;
    if (seqlens_k_.has_value()) {
        const auto& seqlens_k = seqlens_k_.value();
        MFA_CHECK_SAME_DEVICE(seqlens_k, q);
        TORCH_CHECK(seqlens_k.scalar_type() == at::kInt && seqlens_k.numel() == batch && seqlens_k.is_contiguous(),
                    "seqlens_k must be a contiguous int32 tensor with one element per batch entry");
        seqlens_ptr = seqlens_k.data_ptr<int>();
    }
    // ---- append the new tokens first (flash-attn semantics: the cache is updated in place) ----
    int appended = 0;
    if (k_new_.has_value() || v_new_.has_value()) {
        TORCH_CHECK(k_new_.has_value() && v_new_.has_value(), "k and v must be given together");
        TORCH_CHECK(seqlens_ptr != nullptr, "cache_seqlens is required when appending k, v");
        const auto &kn = k_new_.value(), &vn = v_new_.value();
        TORCH_CHECK(kn.scalar_type() == q.scalar_type() && vn.scalar_type() == q.scalar_type(), "k, v dtype must match q");
        MFA_CHECK_SAME_DEVICE(kn, q);
        MFA_CHECK_SAME_DEVICE(vn, q);
        TORCH_CHECK(kn.dim() == 4 && kn.stride(-1) == 1 && vn.stride(-1) == 1, "k, v must be (batch, seqlen_new, heads_k, head_dim)");
        appended = kn.size(1);
        MFA_CHECK_SHAPE(kn, batch, appended, kv_num_heads, head_dim);
        MFA_CHECK_SHAPE(vn, batch, appended, kv_num_heads, head_dim);
        mfa_kvcache_append_params ap{};
        ap.k_new = kn.data_ptr(); ap.v_new = vn.data_ptr();
        ap.k_cache = k_cache.data_ptr(); ap.v_cache = v_cache.data_ptr();
        ap.kn_batch_stride = kn.stride(0); ap.kn_row_stride = kn.stride(1); ap.kn_head_stride = kn.stride(2);
        ap.vn_batch_stride = vn.stride(0); ap.vn_row_stride = vn.stride(1); ap.vn_head_stride = vn.stride(2);
        ap.kc_batch_stride = k_cache.stride(0); ap.kc_row_stride = k_cache.stride(1); ap.kc_head_stride = k_cache.stride(2);
        ap.vc_batch_stride = v_cache.stride(0); ap.vc_row_stride = v_cache.stride(1); ap.vc_head_stride = v_cache.stride(2);
        ap.seqlens_k = seqlens_ptr;
        ap.batch = batch; ap.seqlen_new = appended; ap.kv_heads = kv_num_heads; ap.head_dim = head_dim;
        ap.seqlen_k = seqlen_k;
        if (paged_kv) {
            ap.block_table = p.block_table; ap.block_table_batch_stride = p.block_table_batch_stride;
            ap.page_block_size = p.page_block_size; ap.max_blocks_per_seq = p.max_blocks_per_seq;
        }
        ap.is_bf16 = q.scalar_type() == at::kBFloat16;
        check_rc(mfa_kvcache_append(&ap, current_stream(q)));
    }
    at::Tensor out = at::empty_like(q);
    set_tensor_strides(p, q, k_cache, v_cache, out, true);
    p.batch = batch; p.seqlen_q = seqlen_q; p.seqlen_k = seqlen_k;
    p.heads = num_heads; p.kv_heads = kv_num_heads; p.head_dim = head_dim;
    mfa_forward_params_set_scale(&p);
    p.seqlens_k = seqlens_ptr;
    p.seqlens_k_offset = appended;
    auto opts = q.options().dtype(at::kFloat);
    std::optional<at::Tensor> lse;
    // one entry for every seqlen_q: the library picks flash decoding (Sq = 1, small GQA group), the packed-row MFMA
    // kernel (a few query tokens and / or a large group) or the per-head prefill kernel (long query blocks); the
    // queries are the last seqlen_q positions, so causal / windows align to the last key
    set_windows(p, -1, causal ? 0 : -1, seqlen_k);
    set_extras(p, window_left, window_right, true);
    at::Tensor softmax_lse = at::empty({batch, num_heads, seqlen_q}, opts);
    p.softmax_lse_ptr = softmax_lse.data_ptr<float>();
    run_kvcache(p, q, num_splits);
    if (return_lse) lse = softmax_lse;
    return {out, lse};
}

} // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "mini flash attention (MI355X / gfx950)";
    // positional-only, same order as the reference's csrc/api.cpp:6-8
    m.def("mini_flash_attention_forward", &flash_attention_forward, "Forward pass");
    m.def("mini_flash_attention_varlen_forward", &flash_attention_varlen_forward,
          "Forward pass with variable-length sequences");
    m.def("mini_flash_attention_with_kvcache", &mha_fwd_kvcache, "Forward pass with kv-cache for decoding");
    // opt-in supersets (not in the reference's ABI)
    m.def("forward_ex", &forward_ex, "Forward pass with sliding window / mask alignment / LSE output");
    m.def("varlen_forward_ex", &varlen_forward_ex, "Variable-length forward pass with sliding window / LSE output");
    m.def("kvcache_ex", &kvcache_ex, "KV-cache attention with append, seqlen_q >= 1, sliding window, LSE output");
}
