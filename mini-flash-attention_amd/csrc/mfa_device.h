// Device-side helpers shared by the gfx950 prefill and decode kernels.
// Wave size is 64 on CDNA4; everything here assumes it.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mfa {

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

struct Half {};   // fp16 storage
struct BFloat {}; // bf16 storage

template <typename T>
struct Elem;

template <>
struct Elem<Half> {
    typedef f16x8 frag8;
    // two packed 16-bit values -> two floats
    static __device__ __forceinline__ float lo(uint32_t w) {
        return (float)__builtin_bit_cast(f16x2, w)[0];
    }
    static __device__ __forceinline__ float hi(uint32_t w) {
        return (float)__builtin_bit_cast(f16x2, w)[1];
    }
    static __device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
        return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, a), __builtin_bit_cast(f16x2, b), c, false);
    }
    // round-to-nearest-even pack of two floats
    static __device__ __forceinline__ uint32_t pack(float a, float b) {
        f16x2 r;
        r[0] = (_Float16)a;
        r[1] = (_Float16)b;
        return __builtin_bit_cast(uint32_t, r);
    }
    static __device__ __forceinline__ f32x16 mfma32(frag8 a, frag8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};

template <>
struct Elem<BFloat> {
    typedef bf16x8 frag8;
    static __device__ __forceinline__ float lo(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
    static __device__ __forceinline__ float hi(uint32_t w) {
        return __builtin_bit_cast(float, w & 0xffff0000u);
    }
    static __device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
        return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a), __builtin_bit_cast(bf16x2, b), c,
                                               false);
    }
    static __device__ __forceinline__ uint32_t pack(float a, float b) {
        bf16x2 r;
        r[0] = (__bf16)a;
        r[1] = (__bf16)b;
        return __builtin_bit_cast(uint32_t, r);
    }
    static __device__ __forceinline__ f32x16 mfma32(frag8 a, frag8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};

// ---- cross-lane (wave64) -------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}

// hipcc (ROCm 7.2, clang 22) pitfall: `__builtin_bit_cast(float, r[1])` applied DIRECTLY to an element of
// the 2-vector a permlane swap builtin returns reads element 0 (the IR shows `extractvalue ..., 0` twice).
// Always copy the elements into scalars first, as swap_pair() does.
struct SwapPair {
    float a, b; // a = new vdst, b = new src
};
template <bool K32>
__device__ __forceinline__ SwapPair swap_pair(float x) {
    const uint32_t u = __builtin_bit_cast(uint32_t, x);
    uint32_t r0, r1;
    if constexpr (K32) {
        auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        r0 = r[0];
        r1 = r[1];
    } else {
        auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
        r0 = r[0];
        r1 = r[1];
    }
    return SwapPair{__builtin_bit_cast(float, r0), __builtin_bit_cast(float, r1)};
}

// Sum over aligned groups of LANES consecutive lanes (LANES in {4,8,16,32,64}); every lane of the
// group ends with the total.  DPP inside a 16-lane row, permlane swaps across rows.
template <int LANES>
__device__ __forceinline__ float group_sum(float x) {
    x += dpp_mov<0xB1>(x); // quad_perm [1,0,3,2]
    x += dpp_mov<0x4E>(x); // quad_perm [2,3,0,1]
    if constexpr (LANES >= 8) x += dpp_mov<0x141>(x);  // row_half_mirror
    if constexpr (LANES >= 16) x += dpp_mov<0x140>(x); // row_mirror
    if constexpr (LANES >= 32) {
        // {x0,x0,x2,x2} + {x1,x1,x3,x3} (rows of 16 lanes)
        const SwapPair r = swap_pair<false>(x);
        x = r.a + r.b;
    }
    if constexpr (LANES >= 64) {
        const SwapPair r = swap_pair<true>(x);
        x = r.a + r.b;
    }
    return x;
}

// value held by lane (lane ^ 32)
__device__ __forceinline__ float swap32(float x) {
    const SwapPair r = swap_pair<true>(x);
    // r.a = {x[0:31], x[0:31]}, r.b = {x[32:63], x[32:63]}
    return (threadIdx.x & 32) ? r.a : r.b;
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// ---- LDS-DMA: 16 bytes per lane from global memory straight into LDS (no VGPR destination) -----------------
// The wave's 64 lanes land at lds_addr + 16*lane (lds_addr wave-uniform, in M0); the SOURCE address is per lane.
// Issued through inline asm on purpose: with __builtin_amdgcn_global_load_lds hipcc (ROCm 7.2) treats the DMA as
// a store that may alias every later LDS read and drains it with s_waitcnt vmcnt(0) before the first
// ds_read_b64_tr_b16 of the tile, i.e. half a tile after it was issued.  In asm the compiler does not track it:
// the CALLER must retire it with s_waitcnt vmcnt(N) and a workgroup barrier before any wave reads the bytes.
// M0 is written in the same statement that consumes it and is not used by anything else in these kernels.
// NT = the non-temporal cache policy: for data read once (a K/V cache being streamed) it lifts the achievable HBM
// rate from 5.8-6.0 to 6.8-6.9 TB/s (tools/probes/stream_probe.hip); for tiles that other workgroups re-read out of
// L2 (prefill) the default policy is the right one.
template <bool NT = false>
__device__ __forceinline__ void lds_dma16(const char* wave_uniform_base, uint32_t lane_byte_offset, uint32_t lds_addr) {
    if constexpr (NT)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt"
                     :
                     : "v"(lane_byte_offset), "s"(wave_uniform_base), "s"(lds_addr)
                     : "memory");
    else
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                     :
                     : "v"(lane_byte_offset), "s"(wave_uniform_base), "s"(lds_addr)
                     : "memory");
}
template <bool NT = false>
__device__ __forceinline__ void lds_dma16(const char* lane_ptr, uint32_t lds_addr) {
    if constexpr (NT)
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" : : "v"(lane_ptr), "s"(lds_addr) : "memory");
    else
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(lane_ptr), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ uint32_t lds_address(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}

} // namespace mfa
