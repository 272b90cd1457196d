// KV-cache append for gfx950: copy the new K/V rows of a decode / speculative step into the (dense or paged) cache.
// Pure HBM byte movement, 16 bytes per lane, one thread per 16-byte chunk of one (batch, new row, kv head).
// Not in the reference (promised by mini_flash_attention/interface.py:110-111, done in Python by its tests,
// tests/test_flash_decoding.py:574-597); semantics follow flash-attn's flash_attn_with_kvcache(k=, v=).
#include "mfa_device.h"
#include "mfa_launch.h"

namespace mfa {

__global__ __launch_bounds__(256) void kvcache_append_kernel(const mfa_kvcache_append_params p) {
    const int ch = p.head_dim / 8; // 16-byte chunks per row
    const int64_t total = (int64_t)p.batch * p.seqlen_new * p.kv_heads * ch;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int c = (int)(idx % ch);
        int64_t t = idx / ch;
        const int hk = (int)(t % p.kv_heads);
        t /= p.kv_heads;
        const int i = (int)(t % p.seqlen_new);
        const int b = (int)(t / p.seqlen_new);
        const int pos = (p.seqlens_k ? p.seqlens_k[b] : 0) + i;
        if (pos < 0 || pos >= p.seqlen_k) continue; // past the cache capacity: dropped
        int64_t kdst, vdst;
        if (p.block_table) {
            const int pg = pos / p.page_block_size, in = pos - pg * p.page_block_size;
            const int64_t pid = p.block_table[(int64_t)b * p.block_table_batch_stride + pg];
            kdst = pid * p.kc_batch_stride + (int64_t)in * p.kc_row_stride;
            vdst = pid * p.vc_batch_stride + (int64_t)in * p.vc_row_stride;
        } else {
            kdst = (int64_t)b * p.kc_batch_stride + (int64_t)pos * p.kc_row_stride;
            vdst = (int64_t)b * p.vc_batch_stride + (int64_t)pos * p.vc_row_stride;
        }
        const int64_t ksrc = (int64_t)b * p.kn_batch_stride + (int64_t)i * p.kn_row_stride + (int64_t)hk * p.kn_head_stride;
        const int64_t vsrc = (int64_t)b * p.vn_batch_stride + (int64_t)i * p.vn_row_stride + (int64_t)hk * p.vn_head_stride;
        const u32x4 kv = *(const u32x4*)((const char*)p.k_new + 2 * ksrc + 16 * c);
        const u32x4 vv = *(const u32x4*)((const char*)p.v_new + 2 * vsrc + 16 * c);
        *(u32x4*)((char*)p.k_cache + 2 * (kdst + (int64_t)hk * p.kc_head_stride) + 16 * c) = kv;
        *(u32x4*)((char*)p.v_cache + 2 * (vdst + (int64_t)hk * p.vc_head_stride) + 16 * c) = vv;
    }
}

int launch_kvcache_append(const mfa_kvcache_append_params& p, hipStream_t stream) {
    const int64_t total = (int64_t)p.batch * p.seqlen_new * p.kv_heads * (p.head_dim / 8);
    if (total <= 0) return 0;
    const int64_t blocks = (total + 255) / 256;
    const unsigned grid = (unsigned)(blocks < 2048 ? blocks : 2048);
    hipLaunchKernelGGL(kvcache_append_kernel, dim3(grid), dim3(256), 0, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

} // namespace mfa
