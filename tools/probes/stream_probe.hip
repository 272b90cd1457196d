// Streaming-read probe: how fast can a CU pull a big buffer through (a) global_load_dwordx4 into registers,
// (b) global_load_lds_dwordx4 (LDS-DMA), with 2 or 4 KiB-per-wave chunks in flight?  (developer tool)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/stream_probe.hip -o /tmp/stream_probe && /tmp/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void lds_dma16(const char* base, uint32_t off, uint32_t lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(off), "s"(base), "s"(lds) : "memory");
}
__device__ __forceinline__ void lds_dma16_nt(const char* base, uint32_t off, uint32_t lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" : : "v"(off), "s"(base), "s"(lds) : "memory");
}

// each workgroup (256 threads) streams `bytes_per_wg` contiguous bytes; DEPTH = 1-KiB pieces per wave in flight
template <int MODE, int DEPTH>
__global__ __launch_bounds__(256, 2) void stream_kernel(const char* src, size_t bytes_per_wg, uint32_t* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char* base = src + (size_t)blockIdx.x * bytes_per_wg;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * DEPTH * 2 * 1024;
    const size_t per_wave = bytes_per_wg / 4;
    const char* wbase = base + wave * per_wave;
    const int steps = (int)(per_wave / (DEPTH * 1024));
    uint32_t acc = 0;
    if (MODE == 0 || MODE == 3) {
        for (int s = 0; s < steps; ++s) {
            u32x4 v[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const u32x4* ptr = (const u32x4*)(wbase + ((size_t)s * DEPTH + d) * 1024 + lane * 16);
                v[d] = MODE == 0 ? __builtin_nontemporal_load(ptr) : *ptr;
            }
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) acc ^= v[d][0] ^ v[d][1] ^ v[d][2] ^ v[d][3];
        }
    } else {
        // double-buffered: issue DEPTH pieces into buffer p, wait for the previous DEPTH pieces, read one word of them
        for (int s = 0; s < steps; ++s) {
            const int p = s & 1;
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const uint32_t lds = __builtin_amdgcn_readfirstlane(lds0 + (p * DEPTH + d) * 1024);
                if (MODE == 2) lds_dma16_nt(wbase + ((size_t)s * DEPTH + d) * 1024, lane * 16, lds);
                else lds_dma16(wbase + ((size_t)s * DEPTH + d) * 1024, lane * 16, lds);
            }
            // previous step's pieces: all but the DEPTH just issued
            if (DEPTH == 2) __builtin_amdgcn_s_waitcnt(0x0F72);
            else if (DEPTH == 4) __builtin_amdgcn_s_waitcnt(0x0F74);
            else __builtin_amdgcn_s_waitcnt(0x0F78);
            if (s > 0) acc ^= *(const uint32_t*)(smem + wave * DEPTH * 2 * 1024 + ((p ^ 1) * DEPTH) * 1024 + lane * 4);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE, int DEPTH>
static void run(const char* name, const char* d, size_t total, uint32_t* sink, int wgs) {
    const size_t per = total / wgs / 8192 * 8192;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const size_t smem = (MODE == 1 || MODE == 2) ? 4 * DEPTH * 2 * 1024 : 0;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((stream_kernel<MODE, DEPTH>), dim3(wgs), dim3(256), smem, 0, d, per, sink);
    hipEventRecord(a);
    const int it = 10;
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL((stream_kernel<MODE, DEPTH>), dim3(wgs), dim3(256), smem, 0, d, per, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-34s wgs=%5d  %7.1f us  %6.0f GB/s\n", name, wgs, ms / it * 1e3, (double)per * wgs / (ms / it * 1e-3) / 1e9);
}

int main() {
    const size_t total = 1ull << 30; // 1 GiB > 256 MiB Infinity Cache
    char* d; uint32_t* sink;
    hipMalloc(&d, total); hipMalloc(&sink, 4);
    hipMemset(d, 1, total);
    for (int wgs : {256, 512, 1024, 2048}) {
        run<0, 4>("global_load x4 nt (regs), depth 4", d, total, sink, wgs);
        run<0, 8>("global_load x4 nt (regs), depth 8", d, total, sink, wgs);
        run<3, 8>("global_load x4 (regs), depth 8", d, total, sink, wgs);
        run<2, 4>("LDS-DMA x4 nt, depth 4+4", d, total, sink, wgs);
        run<1, 2>("LDS-DMA x4, depth 2+2", d, total, sink, wgs);
        run<1, 4>("LDS-DMA x4, depth 4+4", d, total, sink, wgs);
        run<1, 8>("LDS-DMA x4, depth 8+8", d, total, sink, wgs);
    }
    return 0;
}
