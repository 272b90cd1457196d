// Flash-decoding (seqlen_q == 1) for gfx950: a vector x matrix path, HBM-bound, no MFMA, no LDS staging.
//
// Replaces flash_attention_fwd_split_kv_kernel / _combine_kernel of the reference
// (csrc/mfa/decode.cuh:524-662, :666-755) with a different decomposition:
//   * one workgroup per (batch, KV head, split): all G = Hq/Hkv query heads of the KV head are kept in
//     registers, so every K/V byte is streamed from HBM exactly once (the reference launches one block
//     per QUERY head and re-streams the KV head G times, flash.cu:45);
//   * K/V rows go straight from global memory to VGPRs as 16-byte chunks (LPR lanes cover one row of D
//     elements, a wave-instruction covers 64/LPR rows), two register buffers deep, so each wave keeps
//     2*UNR KiB in flight; nothing is staged through LDS because nothing is reused;
//   * every group of LPR lanes owns its keys' online-softmax state (m, l, acc): the main loop has no
//     traffic between lane groups; the groups of a wave merge by permlane swaps and the 4 waves through
//     LDS once, after the loop;
//   * pages are resolved per KEY (block_table lookups one iteration ahead), so any page_block_size is
//     correct (the reference resolves once per 64-key tile: decode.cuh:50-55).
// Numerics follow decode.cuh:296-312 (fp32 dot), :367-383 (online softmax in the log2 domain, P kept
// fp32), :587-661 (merge, LSE = M*scale + ln L), :718-747 (combine, here max-subtracted).
#include <cstdlib>

#include "mfa_device.h"
#include "mfa_launch.h"
#include "mfa_combine.h"

namespace mfa {

struct DecodeArgs {
    const void* q;
    const void* k;
    const void* v;
    void* o;
    float* lse;      // (B,H) or null
    float* lse_acc;  // (S,B,H)
    float* o_acc;    // (S,B,H,D)
    int32_t* split_ctr; // one arrival counter per (batch, KV head, head chunk), zero between launches: the last split to
                        // arrive merges the partials itself; null: decode_combine_kernel is launched behind
    const int32_t* seqlens_k;
    const int32_t* block_table;
    int64_t q_batch_stride, q_head_stride;
    int64_t o_batch_stride, o_head_stride;
    int64_t k_batch_stride, k_head_stride, k_row_stride;
    int64_t v_batch_stride, v_head_stride, v_row_stride;
    int64_t k_block_stride, v_block_stride, table_batch_stride;
    int32_t batch, heads, kv_heads, group, head_dim, seqlen_k;
    int32_t page_size, page_shift; // page_shift >= 0 when page_size is a power of two
    int32_t max_blocks;
    int32_t num_splits, nchunks;
    int32_t seqlens_k_offset; // added to seqlens_k[b] (rows appended just before this launch)
    float scale_log2; // softmax_scale * log2(e)
    // combine only: query positions per batch entry (1 for flash decoding; the packed-row kv-cache kernels of
    // mfa_prefill.hip write (S,B,Sq,H) partials) and the row stride of O
    int32_t seqlen_q;
    int64_t o_row_stride;
};

constexpr int kDecodeThreads = 256;
constexpr int kDecodeWaves = 4;
constexpr int kUnroll = 4; // 16-byte loads of K (and of V) per lane per iteration

template <typename T, int LPR, int GT>
struct DecodeState {
    float m[GT];      // running max of scaled (log2-domain) scores
    float l[GT];      // running sum of exp2
    float acc[GT][8]; // this lane's 8 output columns
};

template <int N>
struct RowBuf {
    u32x4 r[N];
};

// kPagedAligned: page_size is a power of two and a multiple of the workgroup iteration (TILE keys, TILE <= 64): a whole
// iteration lies in one page, so the page is ONE scalar block-table load per iteration and the rows are addressed as in
// the dense case, 32-bit lane offsets on a wave-uniform base (the reference resolves a page per 64-key tile the same way,
// decode.cuh:50-55); the other paged modes look a page up per key on the VALU.
enum { kDense = 0, kPagedPow2 = 1, kPagedDiv = 2, kPagedAligned = 3 };

template <typename T, int LPR, int GT, int MODE>
__global__ __launch_bounds__(kDecodeThreads) void decode_split_kv_kernel(const DecodeArgs a) {
    constexpr int RPL = 64 / LPR;                          // rows per wave-instruction
    constexpr int TILE = kDecodeWaves * kUnroll * RPL;     // keys per workgroup iteration
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane % LPR; // 16-byte chunk of the row
    const int lg = lane / LPR;
    const bool col_ok = c * 8 < a.head_dim;

    // workgroup -> (row = (batch, KV head, head chunk), split): workgroups bid, bid + 8, ... share an XCD (round-robin dispatch;
    // mfa_init() checks the premise), and row r runs on XCD r & 7 with ALL its splits -- the partials of a row meet in one L2,
    // which the in-kernel merge below relies on -- while neighbouring rows (the KV heads of one batch element) sit on different
    // XCDs, each reading its 256-byte piece of the same K/V rows: unsplit this is the plain order bid = row.  (Round 2 gave XCD
    // x a CONTIGUOUS range of rows: 3 % slower on config 3, 5 % on the README MHA shape, same box, profiles/r03a_ab_*.)
    // A small row count that is no multiple of 8 (fewer than 8: one or two long sequences on a tensor-parallel shard's one or two
    // KV heads) with key splits: one XCD per row would load the XCDs unevenly or leave some idle (spread_splits, mfa_launch.h) --
    // the splits go out in plain order over all of them, and the merge is the separate launch (launch_decode hands over no
    // counters then).
    int split, row;
    {
        const int nrows = a.batch * a.kv_heads * a.nchunks;
        if (spread_splits(nrows, a.num_splits)) {
            split = blockIdx.x / nrows;
            row = blockIdx.x - split * nrows;
        } else {
            const int x = blockIdx.x & 7, k = blockIdx.x >> 3;
            const int ri = k / a.num_splits;
            split = k - ri * a.num_splits;
            row = 8 * ri + x;
            if (row >= nrows) return;
        }
    }
    const int b = row / (a.kv_heads * a.nchunks);
    const int hk = (row - b * a.kv_heads * a.nchunks) / a.nchunks;
    const int chunk = row - (b * a.kv_heads + hk) * a.nchunks;
    const int g0 = chunk * GT; // first query head (within the group) of this workgroup

    int len = a.seqlens_k ? a.seqlens_k[b] + a.seqlens_k_offset : a.seqlen_k;
    len = min(max(len, 0), a.seqlen_k);
    // split ranges in units of 64-key tiles (decode.cuh:26-30)
    const int ntiles = (len + 63) >> 6;
    const int per = (ntiles + a.num_splits - 1) / a.num_splits;
    const int kbeg = min(split * per, ntiles) << 6;
    const int kend = min(min((split + 1) * per, ntiles) << 6, len);

    // ---- query fragments: q[g] = 8 elements of this lane's chunk, kept packed -----------------
    uint32_t qf[GT][4];
#pragma unroll
    for (int g = 0; g < GT; ++g) {
        const int gq = min(g0 + g, a.group - 1);
        const char* qp = (const char*)a.q + 2 * (b * a.q_batch_stride + (int64_t)(hk * a.group + gq) * a.q_head_stride);
        const u32x4 t = *(const u32x4*)(qp + (col_ok ? 16 * c : 0));
#pragma unroll
        for (int i = 0; i < 4; ++i) qf[g][i] = col_ok ? t[i] : 0u;
    }

    DecodeState<T, LPR, GT> st;
#pragma unroll
    for (int g = 0; g < GT; ++g) {
        st.m[g] = -INFINITY;
        st.l[g] = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) st.acc[g][i] = 0.f;
    }

    constexpr bool paged = MODE != kDense;
    const char* kbase = (const char*)a.k + 2 * ((int64_t)hk * a.k_head_stride + (paged ? 0 : b * a.k_batch_stride));
    const char* vbase = (const char*)a.v + 2 * ((int64_t)hk * a.v_head_stride + (paged ? 0 : b * a.v_batch_stride));
    const int32_t* table = paged ? a.block_table + b * a.table_batch_stride : nullptr;
    // lanes whose 16-byte chunk lies past head_dim (D not a multiple of 8*LPR) read chunk 0 and contribute zeros
    const int cc = col_ok ? c : 0;
    const uint32_t cmask = col_ok ? 0xffffffffu : 0u;

    // key handled by (this wave, this lane group) for unroll slot u of the iteration starting at k0
    auto key_of = [&](int k0, int u) { return k0 + (u * kDecodeWaves + wave) * RPL + lg; };

    // Loads never branch: a key past the end of this split is clamped to its last key (a valid, finite row whose
    // score is masked to -inf below), so over-running iterations only re-read one cached row.
    // page id for the clamped key (paged), looked up one iteration ahead of the row loads that need it
    auto page_lookup = [&](int key, int u = 0) -> int {
        if constexpr (!paged) return 0;
        if constexpr (MODE == kPagedAligned) { // one page per iteration (the same for every lane): slot 0 carries it
            if (u != 0) return 0;
            return __builtin_amdgcn_readfirstlane(table[min(min(key, kend - 1) >> a.page_shift, a.max_blocks - 1)]);
        }
        key = min(key, kend - 1);
        const int pg = MODE == kPagedPow2 ? (key >> a.page_shift) : (key / a.page_size);
        return table[min(pg, a.max_blocks - 1)];
    };
    auto load_rows = [&](RowBuf<kUnroll>& kb, RowBuf<kUnroll>& vb, int k0, const int (&pid)[kUnroll]) {
        const char *kpage = kbase, *vpage = vbase;
        if constexpr (MODE == kPagedAligned) {
            const int64_t pb = pid[0]; // looked up one iteration ahead
            kpage = kbase + 2 * pb * a.k_block_stride;
            vpage = vbase + 2 * pb * a.v_block_stride;
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int key = min(key_of(k0, u), kend - 1);
            if constexpr (!paged || MODE == kPagedAligned) {
                // 32-bit byte offsets on a wave-uniform base (one (batch, kv head) slab, or one page, is < 4 GiB)
                const uint32_t row = MODE == kPagedAligned ? (uint32_t)(key & (a.page_size - 1)) : (uint32_t)key;
                const uint32_t ko = row * (uint32_t)(2 * a.k_row_stride) + 16 * cc;
                const uint32_t vo = row * (uint32_t)(2 * a.v_row_stride) + 16 * cc;
                kb.r[u] = __builtin_nontemporal_load((const u32x4*)(kpage + ko));
                vb.r[u] = __builtin_nontemporal_load((const u32x4*)(vpage + vo));
            } else {
                const int in = MODE == kPagedPow2 ? (key & (a.page_size - 1)) : (key % a.page_size);
                const int64_t ko = (int64_t)pid[u] * a.k_block_stride + (int64_t)in * a.k_row_stride;
                const int64_t vo = (int64_t)pid[u] * a.v_block_stride + (int64_t)in * a.v_row_stride;
                kb.r[u] = __builtin_nontemporal_load((const u32x4*)(kbase + 2 * ko + 16 * cc));
                vb.r[u] = __builtin_nontemporal_load((const u32x4*)(vbase + 2 * vo + 16 * cc));
            }
        }
    };

    auto compute = [&](const RowBuf<kUnroll>& kb, const RowBuf<kUnroll>& vb, int k0) {
        float s[kUnroll][GT];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const bool valid = key_of(k0, u) < kend;
#pragma unroll
            for (int g = 0; g < GT; ++g) {
                float d = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) d = Elem<T>::dot2(qf[g][i], kb.r[u][i], d);
                d = group_sum<LPR>(d);
                s[u][g] = valid ? d * a.scale_log2 : -INFINITY;
            }
        }
#pragma unroll
        for (int g = 0; g < GT; ++g) {
            float mx = st.m[g];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) mx = fmaxf(mx, s[u][g]);
            const float ms = (mx == -INFINITY) ? 0.f : mx;
            const float alpha = fast_exp2(st.m[g] - ms);
            st.m[g] = mx;
            float p[kUnroll];
            float ps = 0.f;
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                p[u] = fast_exp2(s[u][g] - ms);
                ps += p[u];
            }
            st.l[g] = st.l[g] * alpha + ps;
            // acc[g] = acc[g]*alpha + sum_u p[u] * v[u]  on register pairs (v_pk_mul_f32 / v_pk_fma_f32)
            typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x2 acc2 = f32x2{st.acc[g][2 * i], st.acc[g][2 * i + 1]} * f32x2{alpha, alpha};
#pragma unroll
                for (int u = 0; u < kUnroll; ++u) {
                    const uint32_t w = vb.r[u][i] & cmask;
                    acc2 = __builtin_elementwise_fma(f32x2{p[u], p[u]}, f32x2{Elem<T>::lo(w), Elem<T>::hi(w)}, acc2);
                }
                st.acc[g][2 * i] = acc2[0];
                st.acc[g][2 * i + 1] = acc2[1];
            }
        }
    };

    if (kbeg < kend) {
        // two register buffers, iterations taken in pairs; every load and compute of the pair is unconditional
        RowBuf<kUnroll> kA, vA, kB, vB;
        int pid[kUnroll], pidn[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) pid[u] = page_lookup(key_of(kbeg, u), u);
        load_rows(kA, vA, kbeg, pid);
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) pid[u] = page_lookup(key_of(kbeg + TILE, u), u);
        // sched_barrier: without it hipcc's scheduler, chasing occupancy, sinks every load down to its first use
        // (load; s_waitcnt vmcnt(0); use), which serialises the whole stream.  The loads of the NEXT buffer must
        // issue before the compute on the CURRENT one.
        for (int k0 = kbeg; k0 < kend; k0 += 2 * TILE) {
            load_rows(kB, vB, k0 + TILE, pid);
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) pidn[u] = page_lookup(key_of(k0 + 2 * TILE, u), u);
            __builtin_amdgcn_sched_barrier(0);
            compute(kA, vA, k0);
            __builtin_amdgcn_sched_barrier(0);
            load_rows(kA, vA, k0 + 2 * TILE, pidn);
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) pid[u] = page_lookup(key_of(k0 + 3 * TILE, u), u);
            __builtin_amdgcn_sched_barrier(0);
            compute(kB, vB, k0 + TILE);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- merge the lane groups of this wave (xor LPR, 2*LPR, ... 32) ---------------------------
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1) {
#pragma unroll
        for (int g = 0; g < GT; ++g) {
            const float mo = __shfl_xor(st.m[g], off);
            const float lo = __shfl_xor(st.l[g], off);
            const float mn = fmaxf(st.m[g], mo);
            const float ms = (mn == -INFINITY) ? 0.f : mn;
            const float fa = fast_exp2(st.m[g] - ms), fb = fast_exp2(mo - ms);
            st.m[g] = mn;
            st.l[g] = st.l[g] * fa + lo * fb;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float ao = __shfl_xor(st.acc[g][i], off);
                st.acc[g][i] = st.acc[g][i] * fa + ao * fb;
            }
        }
    }

    // ---- merge the 4 waves through LDS ------------------------------------------------------
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sm_ml = smem;                              // [wave][GT][2]
    float* sm_acc = smem + kDecodeWaves * GT * 2;     // [wave][GT][LPR*8]
    if (lane < LPR) {
#pragma unroll
        for (int g = 0; g < GT; ++g) {
            if (lane == 0) {
                sm_ml[(wave * GT + g) * 2 + 0] = st.m[g];
                sm_ml[(wave * GT + g) * 2 + 1] = st.l[g];
            }
            float* dst = sm_acc + ((wave * GT + g) * LPR + c) * 8;
            *(f32x4*)(dst) = f32x4{st.acc[g][0], st.acc[g][1], st.acc[g][2], st.acc[g][3]};
            *(f32x4*)(dst + 4) = f32x4{st.acc[g][4], st.acc[g][5], st.acc[g][6], st.acc[g][7]};
        }
    }
    __syncthreads();

    const int D = a.head_dim;
    for (int idx = threadIdx.x; idx < GT * D; idx += kDecodeThreads) {
        const int g = idx / D, d = idx - g * D;
        if (g0 + g >= a.group) break;
        float M = -INFINITY;
#pragma unroll
        for (int w = 0; w < kDecodeWaves; ++w) M = fmaxf(M, sm_ml[(w * GT + g) * 2]);
        const float Ms = (M == -INFINITY) ? 0.f : M;
        float L = 0.f, o = 0.f;
#pragma unroll
        for (int w = 0; w < kDecodeWaves; ++w) {
            const float f = fast_exp2(sm_ml[(w * GT + g) * 2] - Ms);
            L += sm_ml[(w * GT + g) * 2 + 1] * f;
            o += sm_acc[(w * GT + g) * LPR * 8 + d] * f;
        }
        const float inv = L > 0.f ? 1.f / L : 0.f;
        o *= inv;
        const int hq = hk * a.group + g0 + g;
        // natural-log LSE of the scaled scores: M is in the log2 domain
        const float lse = L > 0.f ? M * 0.6931471805599453f + __logf(L) : -INFINITY;
        if (a.num_splits > 1) {
            const int64_t slot = ((int64_t)split * a.batch + b) * a.heads + hq;
            a.o_acc[slot * D + d] = o;
            if (d == 0) a.lse_acc[slot] = lse;
        } else {
            char* op = (char*)a.o + 2 * (b * a.o_batch_stride + (int64_t)hq * a.o_head_stride + d);
            const uint32_t pk = Elem<T>::pack(o, 0.f);
            *(uint16_t*)op = (uint16_t)pk;
            if (d == 0 && a.lse) a.lse[(int64_t)b * a.heads + hq] = lse;
        }
    }
    if (a.num_splits > 1 && a.split_ctr) {
        // the last split of this row to arrive merges the partials (as the packed-row kernels do, mfa_prefill.hip): stores
        // drained into the XCD's L2, one ticket per workgroup, acquire and counter reset in the winner, one wave per head
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int* const flag = (int*)smem;
        if (threadIdx.x == 0) *flag = atomicAdd(a.split_ctr + row, 1);
        __syncthreads();
        if (*flag != a.num_splits - 1) return;
        // (no acquire fence: the winner reads the partials with L2-served loads, combine_row<T, true>)
        if (threadIdx.x == 0) a.split_ctr[row] = 0;
        const int64_t BH = (int64_t)a.batch * a.heads;
        for (int g = wave; g < GT && g0 + g < a.group; g += kDecodeWaves) {
            const int hq = hk * a.group + g0 + g;
            const int64_t bh = (int64_t)b * a.heads + hq;
            char* orow = (char*)a.o + 2 * (b * a.o_batch_stride + (int64_t)hq * a.o_head_stride);
            combine_row<T, true>(a.o_acc, a.lse_acc, a.num_splits, BH, bh, D, orow, a.lse ? a.lse + bh : nullptr, lane);
        }
    }
}

// The split merge as its own launch: one WAVE per (batch, query position, head) row, four rows per workgroup
// (combine_row, mfa_combine.h).  (The first version used a workgroup per row with four barriers: 6.0 us for 1 536
// rows x 8 splits; this one 3 us.)
constexpr int kCombineRows = 4;
template <typename T>
__global__ __launch_bounds__(64 * kCombineRows) void decode_combine_kernel(const DecodeArgs a) {
    const int64_t BH = (int64_t)a.batch * a.seqlen_q * a.heads;
    const int lane = threadIdx.x & 63;
    const int64_t bh = (int64_t)blockIdx.x * kCombineRows + (threadIdx.x >> 6);
    if (bh >= BH) return; // whole wave
    const int h = bh % a.heads;
    const int64_t t = bh / a.heads;
    const int pos = t % a.seqlen_q;
    const int64_t b = t / a.seqlen_q;
    char* orow = (char*)a.o + 2 * (b * a.o_batch_stride + pos * a.o_row_stride + (int64_t)h * a.o_head_stride);
    float* lse_out = a.lse ? a.lse + (b * a.heads + h) * a.seqlen_q + pos : nullptr; // (B, H, Sq)
    combine_row<T>(a.o_acc, a.lse_acc, a.num_splits, BH, bh, a.head_dim, orow, lse_out, lane);
}

template <typename T, int LPR, int GT>
static int launch_decode_t(const DecodeArgs& a, hipStream_t stream) {
    const int64_t R = (int64_t)a.batch * a.kv_heads * a.nchunks;
    dim3 grid((unsigned)(spread_splits(R, a.num_splits) ? R * a.num_splits : 8 * ((R >> 3) + ((R & 7) ? 1 : 0)) * a.num_splits));
    const size_t smem = sizeof(float) * kDecodeWaves * GT * (2 + LPR * 8);
    if (!a.block_table)
        hipLaunchKernelGGL((decode_split_kv_kernel<T, LPR, GT, kDense>), grid, dim3(kDecodeThreads), smem, stream, a);
    else if (a.page_shift >= 0 && LPR >= 16 && a.page_size >= kDecodeWaves * kUnroll * (64 / LPR))
        hipLaunchKernelGGL((decode_split_kv_kernel<T, LPR, GT, kPagedAligned>), grid, dim3(kDecodeThreads), smem, stream, a);
    else if (a.page_shift >= 0)
        hipLaunchKernelGGL((decode_split_kv_kernel<T, LPR, GT, kPagedPow2>), grid, dim3(kDecodeThreads), smem, stream, a);
    else
        hipLaunchKernelGGL((decode_split_kv_kernel<T, LPR, GT, kPagedDiv>), grid, dim3(kDecodeThreads), smem, stream, a);
    if (a.num_splits > 1 && !a.split_ctr) {
        const int64_t BH = (int64_t)a.batch * a.heads;
        hipLaunchKernelGGL((decode_combine_kernel<T>), dim3((unsigned)((BH + kCombineRows - 1) / kCombineRows)), dim3(64 * kCombineRows), 0, stream, a);
    }
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

template <typename T, int LPR>
static int launch_decode_g(const DecodeArgs& a, int gt, hipStream_t stream) {
    switch (gt) {
    case 1: return launch_decode_t<T, LPR, 1>(a, stream);
    case 2: return launch_decode_t<T, LPR, 2>(a, stream);
    case 3: return launch_decode_t<T, LPR, 3>(a, stream);
    case 4: return launch_decode_t<T, LPR, 4>(a, stream);
    case 6: return launch_decode_t<T, LPR, 6>(a, stream);
    default: return launch_decode_t<T, LPR, 8>(a, stream);
    }
}

template <typename T>
static int launch_decode_d(const DecodeArgs& a, int gt, hipStream_t stream) {
    if (a.head_dim <= 32) return launch_decode_g<T, 4>(a, gt, stream);
    if (a.head_dim <= 64) return launch_decode_g<T, 8>(a, gt, stream);
    if (a.head_dim <= 128) return launch_decode_g<T, 16>(a, gt, stream);
    return launch_decode_g<T, 32>(a, gt, stream);
}

// Host launcher: picks the group tile GT (query heads per workgroup) and chunks larger groups.
int launch_decode(const mfa_forward_params& p, hipStream_t stream, bool* merged_in_kernel) {
    DecodeArgs a{};
    a.q = p.q_ptr; a.k = p.k_ptr; a.v = p.v_ptr; a.o = p.o_ptr;
    a.lse = p.softmax_lse_ptr; a.lse_acc = p.softmax_lseaccum_ptr; a.o_acc = p.oaccum_ptr;
    a.seqlens_k = p.seqlens_k; a.block_table = p.block_table;
    a.q_batch_stride = p.q_batch_stride; a.q_head_stride = p.q_head_stride;
    a.o_batch_stride = p.o_batch_stride; a.o_head_stride = p.o_head_stride;
    a.k_batch_stride = p.k_batch_stride; a.k_head_stride = p.k_head_stride; a.k_row_stride = p.k_row_stride;
    a.v_batch_stride = p.v_batch_stride; a.v_head_stride = p.v_head_stride; a.v_row_stride = p.v_row_stride;
    a.k_block_stride = p.k_cache_block_stride; a.v_block_stride = p.v_cache_block_stride;
    a.table_batch_stride = p.block_table_batch_stride;
    a.batch = p.batch; a.heads = p.heads; a.kv_heads = p.kv_heads; a.group = p.heads / p.kv_heads;
    a.head_dim = p.head_dim; a.seqlen_k = p.seqlen_k;
    a.page_size = p.page_block_size > 0 ? p.page_block_size : 1;
    a.page_shift = (a.page_size & (a.page_size - 1)) == 0 ? __builtin_ctz(a.page_size) : -1;
    a.max_blocks = p.max_blocks_per_seq > 0 ? p.max_blocks_per_seq : (p.seqlen_k + a.page_size - 1) / a.page_size;
    a.num_splits = p.num_splits < 1 ? 1 : p.num_splits;
    a.scale_log2 = p.softmax_scale_log2;
    a.seqlens_k_offset = p.seqlens_k_offset;
    a.seqlen_q = 1;
    a.o_row_stride = 0;
    const int G = a.group;
    const int env_gtmax = g_knobs.decode_gt_max.load();
    const int gtmax = env_gtmax > 0 ? env_gtmax : 8;
    int gt = G <= 4 ? G : (G <= 6 ? 6 : 8);
    if (G == 5) gt = 6;
    if (gt > gtmax) { // split the group evenly over ceil(G / gtmax) workgroups
        const int nch = (G + gtmax - 1) / gtmax;
        gt = (G + nch - 1) / nch;
        if (gt == 5) gt = 6;
        if (gt == 7) gt = 8;
    }
    a.nchunks = (G + gt - 1) / gt;
    if ((int64_t)a.batch * a.kv_heads * a.nchunks * a.num_splits >= (1LL << 30)) return -4; // (1-D grid)
    a.split_ctr = nullptr;
    if (a.num_splits > 1) {
        const int64_t rows = (int64_t)a.batch * a.kv_heads * a.nchunks;
        mfa_forward_params one = p; // (the partials of flash decoding: one query position)
        one.seqlen_q = 1;
        one.num_splits = a.num_splits;
        a.split_ctr = pick_split_counters(p, (size_t)rows, rows, rows * a.num_splits, partial_bytes(one));
    }
    if (merged_in_kernel) *merged_in_kernel = a.split_ctr != nullptr;
    return p.is_bf16 ? launch_decode_d<BFloat>(a, gt, stream) : launch_decode_d<Half>(a, gt, stream);
}

// LSE-weighted merge of (S, B, Sq, H[, D]) partials into O / LSE, for the packed-row kv-cache kernels
int launch_decode_combine(const mfa_forward_params& p, hipStream_t stream) {
    DecodeArgs a{};
    a.o = p.o_ptr; a.lse = p.softmax_lse_ptr; a.lse_acc = p.softmax_lseaccum_ptr; a.o_acc = p.oaccum_ptr;
    a.o_batch_stride = p.o_batch_stride; a.o_head_stride = p.o_head_stride; a.o_row_stride = p.o_row_stride;
    a.batch = p.batch; a.heads = p.heads; a.head_dim = p.head_dim; a.num_splits = p.num_splits;
    a.seqlen_q = p.seqlen_q;
    const int64_t rows = (int64_t)p.batch * p.seqlen_q * p.heads;
    if (rows <= 0) return 0;
    const dim3 grid((unsigned)((rows + kCombineRows - 1) / kCombineRows)), block(64 * kCombineRows);
    if (p.is_bf16) hipLaunchKernelGGL((decode_combine_kernel<BFloat>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((decode_combine_kernel<Half>), grid, block, 0, stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

} // namespace mfa
