// The LSE-weighted merge of key-split partials, one output row per WAVE (reference decode.cuh:718-747, here
// max-subtracted):  O = sum_s o_s * exp(lse_s - LSE),  LSE = ln sum_s exp(lse_s).
// Shared by decode_combine_kernel (mfa_decode.hip: its own launch behind flash decoding) and by the packed-row
// kv-cache kernels (mfa_prefill.hip), whose last-arriving key split of a row block runs it in its own epilogue.
#pragma once

#include "mfa_device.h"

namespace mfa {

// o_acc: (S, BH, D) fp32 normalised partials, lse_acc: (S, BH) natural-log LSE (-inf: an empty split), row bh.
// The split weights stay in two registers per lane (S <= 128), every reduction is a wave shuffle (no LDS, no barrier),
// and a lane owns column pairs (2*lane, 2*lane+1) + 128*k of the row, so the partial-O reads are 8-byte coalesced
// and all splits' loads of a column pair are independent.  orow: the row of O (16-bit elements); lse_out: may be null.
// L2LOADS: the partials were written by OTHER workgroups of this launch (the in-kernel merge by the last split to arrive): every
// load of them is an agent-scope relaxed atomic load (`global_load ... sc1`: served by the XCD's L2, never by this CU's L1, which
// may hold lines of an earlier launch's partials at the same addresses).  With every split of a row on one XCD (the launch's
// grid order; mfa_init() checks the premise) and each writer's stores drained (s_waitcnt vmcnt(0)) before its arrival ticket, the
// L2 holds them when the winner's ticket returns -- no L1 invalidate (`buffer_inv sc1`, ~1.7 us per winning workgroup), which is
// what made the in-kernel merge lose on launches of more than one round of workgroups.
template <bool L2LOADS>
__device__ __forceinline__ float ld_f32(const float* p) {
    if constexpr (L2LOADS) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}
template <bool L2LOADS>
__device__ __forceinline__ float2 ld_f32x2(const float* p) {
    if constexpr (L2LOADS) {
        const uint64_t w = __hip_atomic_load((const uint64_t*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return float2{__builtin_bit_cast(float, (uint32_t)w), __builtin_bit_cast(float, (uint32_t)(w >> 32))};
    } else {
        return *(const float2*)p;
    }
}

template <typename T, bool L2LOADS = false>
__device__ __forceinline__ void combine_row(const float* o_acc, const float* lse_acc, int S, int64_t BH, int64_t bh, int D,
                                            char* orow, float* lse_out, int lane) {
    const float l0 = lane < S ? ld_f32<L2LOADS>(lse_acc + lane * BH + bh) : -INFINITY;
    const float l1 = lane + 64 < S ? ld_f32<L2LOADS>(lse_acc + (lane + 64) * BH + bh) : -INFINITY;
    // the first chunk of partial-O loads does not depend on the weights: issue it behind the LSE loads so that the
    // merge pays one memory round trip, not two (it is latency, not bandwidth, that it consists of)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    constexpr int CH = 8;
    const int d0 = 2 * lane;
    const float* src0 = o_acc + bh * D + d0;
    f32x2 first[CH];
#pragma unroll
    for (int u = 0; u < CH; ++u)
        if (u < S && d0 < D) {
            const float2 t = ld_f32x2<L2LOADS>(src0 + (int64_t)u * BH * D);
            first[u] = f32x2{t.x, t.y};
        } else {
            first[u] = f32x2{0.f, 0.f};
        }
    float M = fmaxf(l0, l1);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) M = fmaxf(M, __shfl_xor(M, off));
    const float w0 = (lane < S && M != -INFINITY) ? __expf(l0 - M) : 0.f;
    const float w1 = (lane + 64 < S && M != -INFINITY) ? __expf(l1 - M) : 0.f;
    float W = w0 + w1;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) W += __shfl_xor(W, off);
    const float invW = W > 0.f ? 1.f / W : 0.f;
    // Every lane walks the column loop (the trip count is wave-uniform); only the loads and the store are
    // predicated on d < D.  The split weights travel by v_readlane, which ignores EXEC: a ds_bpermute broadcast
    // under the `d < D` guard returned 0 for weights held by lanes with 2*lane >= D (head dims < 128, splits >= D/2).
    for (int dd = 0; dd < D; dd += 128) { // D is a multiple of 8: pairs never straddle the row end
        const int d = dd + d0;
        const bool col = d < D;
        f32x2 acc = {0.f, 0.f};
        const float* src = o_acc + bh * D + d;
        for (int s0 = 0; s0 < S; s0 += CH) {
            f32x2 v[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                if (dd == 0 && s0 == 0) v[u] = first[u];
                else if (col && s0 + u < S) {
                    const float2 t = ld_f32x2<L2LOADS>(src + (int64_t)(s0 + u) * BH * D);
                    v[u] = f32x2{t.x, t.y};
                } else {
                    v[u] = f32x2{0.f, 0.f};
                }
            }
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int sp = s0 + u; // wave-uniform; weights of splits >= S are 0
                const float w = __builtin_bit_cast(
                    float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sp < 64 ? w0 : w1), sp & 63));
                acc += w * v[u];
            }
        }
        if (col) *(uint32_t*)(orow + 2 * d) = Elem<T>::pack(acc[0] * invW, acc[1] * invW);
    }
    if (lane == 0 && lse_out) *lse_out = (M != -INFINITY) ? M + __logf(W) : -INFINITY;
}

} // namespace mfa
