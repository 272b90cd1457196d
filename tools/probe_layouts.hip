// Empirical probe of the gfx950 cross-lane / MFMA layouts the kernels rely on.  Prints, for each
// primitive, where every lane's data came from, using exact small integers.
//   hipcc --offload-arch=gfx950 -O2 tools/probe_layouts.hip -o gpurun_out/probe && gpurun_out/probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void probe_tr(int* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[64 * 4];
    const int lane = threadIdx.x;
    // element e of the 8 bytes addressed by lane L holds value L*4+e
    for (int e = 0; e < 4; ++e) lds[lane * 4 + e] = lane * 4 + e;
    __syncthreads();
    auto p = (__attribute__((address_space(3))) s16x4*)(lds + lane * 4);
    s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = r[e];
}

__global__ void probe_swap(int* out) {
    const int lane = threadIdx.x;
    auto a = __builtin_amdgcn_permlane16_swap(lane, 100 + lane, false, false);
    auto b = __builtin_amdgcn_permlane32_swap(lane, 100 + lane, false, false);
    out[lane * 4 + 0] = a[0];
    out[lane * 4 + 1] = a[1];
    out[lane * 4 + 2] = b[0];
    out[lane * 4 + 3] = b[1];
}

// C = A.B with A[m][k] = (m == M0 && k == K0), B[k][n] = k*32+n+1  ->  C[M0][n] = K0*32+n+1 tells both maps
__global__ void probe_mfma(float* out, int a_lane, int a_elem, int b_mode) {
    const int lane = threadIdx.x;
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (lane == a_lane && j == a_elem) ? (_Float16)1.0f : (_Float16)0.0f;
        // assumed B map: lane (r,h) element j = B[k = 8h + j][n = r]; encode value k*32+n (exact in fp16 up to 2048)
        const int k = 8 * (lane >> 5) + j, n = lane & 31;
        b[j] = (_Float16)(float)(k * 32 + n + 1);
    }
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    for (int i = 0; i < 16; ++i) out[lane * 16 + i] = c[i];
}

int main() {
    int* d;
    hipMalloc(&d, 64 * 16 * sizeof(float));
    std::vector<int> h(64 * 16);
    probe_tr<<<1, 64>>>(d);
    hipMemcpy(h.data(), d, 64 * 4 * 4, hipMemcpyDeviceToHost);
    printf("ds_read_tr16_b64: lane -> 4 values (value v = source lane v/4, element v%%4)\n");
    for (int l = 0; l < 64; ++l)
        printf("  lane %2d: (%2d,%d) (%2d,%d) (%2d,%d) (%2d,%d)\n", l, h[l * 4] / 4, h[l * 4] % 4, h[l * 4 + 1] / 4,
               h[l * 4 + 1] % 4, h[l * 4 + 2] / 4, h[l * 4 + 2] % 4, h[l * 4 + 3] / 4, h[l * 4 + 3] % 4);
    probe_swap<<<1, 64>>>(d);
    hipMemcpy(h.data(), d, 64 * 4 * 4, hipMemcpyDeviceToHost);
    printf("permlane swaps (vdst_in = lane, src_in = 100+lane): lane: p16[0] p16[1] | p32[0] p32[1]\n");
    for (int l = 0; l < 64; ++l)
        printf("  lane %2d: %3d %3d | %3d %3d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    std::vector<float> f(64 * 16);
    const int tests[][2] = {{0, 0}, {0, 3}, {0, 4}, {5, 1}, {32, 0}, {37, 6}, {63, 7}};
    for (auto& t : tests) {
        probe_mfma<<<1, 64>>>((float*)d, t[0], t[1], 0);
        hipMemcpy(f.data(), d, 64 * 16 * 4, hipMemcpyDeviceToHost);
        printf("mfma 32x32x16 f16: A one-hot at lane %d elem %d -> nonzero C entries (lane,reg)=value-1 -> k=val/32 n=val%%32:\n",
               t[0], t[1]);
        int cnt = 0;
        for (int l = 0; l < 64 && cnt < 6; ++l)
            for (int i = 0; i < 16; ++i)
                if (f[l * 16 + i] != 0.f && cnt < 6) {
                    const int v = (int)f[l * 16 + i] - 1;
                    printf("    C(lane %2d, reg %2d): k=%d n=%d\n", l, i, v / 32, v % 32);
                    ++cnt;
                }
    }
    hipFree(d);
    return 0;
}
