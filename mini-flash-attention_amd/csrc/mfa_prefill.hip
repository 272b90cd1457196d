// FlashAttention-2 forward (prefill / varlen / paged prefill) for gfx950, fp16/bf16 MFMA, wave64.
//
// Replaces flash_attention_fwd_kernel of the reference (csrc/mfa/prefill.cuh:712-803).  Same math
// (Appendix A of SURVEY.md: raw-score running max, exp2 with scale*log2e folded in, fp32 row sums of
// un-rounded P, P rounded to the element type before P.V, fp32 O, 1/l at the end with a 0/NaN guard,
// TOP-LEFT causal mask prefill.cuh:416-419) on a CDNA4-shaped decomposition:
//
//   * workgroup = NW waves, each wave owns 32 query rows (BM = 32*NW); key tile BN = 64;
//   * "swapped" first product  S^T = K . Q^T  with v_mfma_f32_32x32x16: the 32x32 accumulator has the
//     QUERY ROW on the lane and 16 keys in its registers, so row max / row sum are in-lane plus one
//     exchange with lane^32, and no score ever crosses lanes through LDS;
//   * the accumulator is converted in place to 16-bit and used directly as the B operand of the second
//     product  O^T = V^T . P^T  (k-order inside a step is permuted: row 16s + 8(j>>2) + 4h + (j&3));
//     V^T fragments with that same order come from ds_read_b64_tr_b16 transposed LDS reads;
//   * K and V tiles are staged global -> registers -> LDS (issue early, write late), double-buffered,
//     ONE barrier per key tile; K rows are XOR-swizzled per 16-byte chunk for conflict-free
//     ds_read_b128, V rows per 64-byte unit for conflict-free transposed reads;
//   * O leaves through LDS as whole rows (16-byte coalesced stores);
//   * workgroups are numbered so that the query blocks and query heads sharing one (batch, KV head)
//     run on one XCD (shared L2), heaviest causal blocks first.
#include <algorithm>
#include <atomic>
#include <mutex>
#include <cstdlib>
#include <type_traits>

#include "mfa_device.h"
#include "mfa_launch.h"
#include "mfa_combine.h"
#include "mfa_dev.h"
#include "mfa_prefill_args.h"

namespace mfa {


constexpr int kBN = 64; // keys per tile

// LDS row pitch in bytes for a head dim (rows padded to a power of two >= 64 B)
template <int D>
struct Pitch {
    static constexpr int RB = D <= 32 ? 64 : D <= 64 ? 128 : D <= 128 ? 256 : 512;
};

// 16-byte-chunk XOR for the K image (rows read by ds_read_b128, 16 lanes = 16 different rows)
template <int RB>
__device__ __forceinline__ int k_swz(int row) {
    if constexpr (RB == 64) return (row >> 2) & 3;
    else if constexpr (RB == 128) return (row >> 1) & 7;
    else return row & 15;
}
// 64-byte-unit XOR for the V image (transposed reads take 4 consecutive keys x 64 B per half-wave)
template <int RB>
__device__ __forceinline__ int v_swz(int row) {
    if constexpr (RB == 64) return 0;
    else if constexpr (RB == 128) return (row >> 1) & 1;
    else return row & 3;
}

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

// MQ = false: one workgroup per (batch, query head, block of BM query rows): prefill.
// MQ = true : one workgroup per (batch, KV head, key split, block of BM PACKED rows), row = query position * G +
//   head within the group: kv-cache attention with few query positions (speculative / chunked decoding, seqlen_q
//   of 1..a few dozen) or a large GQA group, where one query head alone would leave the 32-row MFMA tiles almost
//   empty and every head would re-stream the same K/V.  Same tile loop; only the row -> (position, head) mapping,
//   the key range (a split) and the destination (final O, or normalised partials for decode_combine_kernel) differ.
// PG: 0 = dense K/V; 1 = paged, any page size (a block-table lookup per staged row); 2 = paged with page_size a power
// of two >= the 64-key tile: a tile lies in one page, one scalar lookup per tile (reference decode.cuh:50-55 resolves
// the same way).
template <typename T, int D, int NW, int PG, bool MQ = false, bool STREAM = false>
__global__ __launch_bounds__(64 * NW, (D <= 128 && !(MQ && PG == 1 && D == 128) ? 2 : 1)) void prefill_fwd_kernel(const PrefillArgs a) {
    constexpr bool PAGED = PG != 0;
    using E = Elem<T>;
    using frag8 = typename E::frag8;
    constexpr int RB = Pitch<D>::RB;
    constexpr int NT = 64 * NW;
    constexpr int BM = 32 * NW;
    constexpr int KS = D / 16;   // k-steps of the first product
    constexpr int DB = D / 32;   // 32-wide output column blocks of the second product
    constexpr int CH = D / 8;    // 16-byte chunks per row
    constexpr int TILE_CHUNKS = kBN * CH;
    constexpr int CPT = (TILE_CHUNKS + NT - 1) / NT; // chunks per thread per tile (K, and V)
    constexpr int TILE_BYTES = kBN * RB;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sK = smem;                  // [2][64][RB]
    char* const sV = smem + 2 * TILE_BYTES; // [2][64][RB]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31;
    const int h = lane >> 5;
    MFA_DEV_STAMP_DECL; // (developer builds only: mfa_dev.h)

    // ---- workgroup -> (batch, head, query block) -------------------------------------------------
    // XCD-aware: blocks bid, bid+8, ... share an XCD (round-robin dispatch), so XCD x = bid & 7 owns a contiguous
    // range of (batch, head) pairs and their K/V stay in that XCD's L2.  Inside an XCD the pairs are taken in
    // groups of `gp`; a group is walked query-block-major (heaviest causal block of each pair first), so that
    // consecutive workgroups carry EQUAL work: the dispatcher deals consecutive workgroups round-robin over the
    // XCD's shader engines, and a heavy-to-light sequence inside one pair lands the heavy blocks on the same
    // engines every time (measured: 70 % wave-slot occupancy with pair-major order vs 94 % non-causal).
    int b, hq, hk, m0, split = 0;
    if constexpr (!MQ) {
        const int nmb = a.num_m_blocks;
        const int npairs = a.batch * a.heads;
        const int x = blockIdx.x & 7, k = blockIdx.x >> 3;
        const int p8 = npairs >> 3, r8 = npairs & 7;
        const int pair_begin = x < r8 ? x * (p8 + 1) : r8 * (p8 + 1) + (x - r8) * p8;
        // per-sequence lengths (varlen, per-batch cache lengths): a contiguous range of pairs per XCD would hand a long
        // sequence's heads to ONE XCD (bf16 H24/8, one 8192-token sequence beside 31 of 256: 2.5 ms, one XCD doing the work of
        // eight).  There the pairs are dealt round-robin: (batch, KV head) pairs with their query heads when there are at least two
        // per XCD (interleave_pairs == 2: XCD x takes x, x + 8, ...; a KV head's K/V stays in one L2), (batch, head) pairs
        // otherwise (== 1)
        const int dealt = a.interleave_pairs == 2 ? a.batch * a.kv_heads : npairs;
        const int mine = (dealt - x + 7) >> 3;
        const int pair_count = a.interleave_pairs == 2 ? mine * a.group : a.interleave_pairs == 1 ? mine : p8 + (x < r8 ? 1 : 0);
        const int gp = a.group_pairs;
        int g = k / (gp * nmb);
        const int t = k - g * (gp * nmb);
        const int gsize = min(gp, pair_count - g * gp); // pairs in this (possibly last, short) group
        int rank, pi;
        if (a.interleave_pairs == 3) { // few pairs, no multiple of 8: no XCD of its own for a pair; query-block-major over all of them
            rank = blockIdx.x / npairs;
            pi = blockIdx.x - rank * npairs;
            g = 0;
        } else {
            if (gsize <= 0) return;
            rank = t / gsize;
            pi = t - rank * gsize;
        }
        if (rank >= nmb) return; // padding of a short group
        const int mblk = nmb - 1 - rank; // heaviest causal blocks first
        if (a.interleave_pairs == 3) {
            hq = pi % a.heads;
            b = pi / a.heads;
            hk = hq / a.group;
        } else if (a.interleave_pairs == 2) {
            const int pl = g * gp + pi, kvl = pl / a.group;
            const int kv = x + 8 * kvl;
            b = kv / a.kv_heads;
            hk = kv - b * a.kv_heads;
            hq = hk * a.group + (pl - kvl * a.group);
        } else if (a.interleave_pairs == 1) {
            const int bh = x + 8 * (g * gp + pi);
            hq = bh % a.heads;
            b = bh / a.heads;
            hk = hq / a.group;
        } else {
            const int bh = pair_begin + g * gp + pi;
            hq = bh % a.heads;
            b = bh / a.heads;
            hk = hq / a.group;
        }
        m0 = mblk * BM;
    } else {
        // (batch, KV head) pair p runs on XCD p & 7 (workgroups bid, bid + 8, ... share one) with all its key splits and row
        // blocks -- the in-kernel merge needs a row's partials in one L2 -- and neighbouring pairs on different XCDs
        const int npairs = a.batch * a.kv_heads;
        const int x = blockIdx.x & 7, k = blockIdx.x >> 3;
        const int per_pair = a.num_splits * a.mq_row_blocks;
        int pi = k / per_pair, t = k - pi * per_pair;
        int bk = 8 * pi + x;
        if (spread_splits(npairs, a.num_splits)) { // (few pairs: their splits and row blocks in plain order over all XCDs, mfa_launch.h)
            t = blockIdx.x / npairs;
            bk = blockIdx.x - t * npairs;
        }
        if (bk >= npairs) return;
        b = bk / a.kv_heads;
        hk = bk - b * a.kv_heads;
        split = t / a.mq_row_blocks;
        m0 = (t - split * a.mq_row_blocks) * BM; // first PACKED row of this block
        hq = hk * a.group;                       // first query head of the group
    }

    int sq, sk;
    int64_t q_off, o_off, k_off, v_off;
    if (!MQ && a.cu_q) {
        const int q0 = a.cu_q[b], k0 = a.cu_k[b];
        sq = a.cu_q[b + 1] - q0;
        sk = a.cu_k[b + 1] - k0;
        q_off = (int64_t)q0 * a.q_row_stride;
        o_off = (int64_t)q0 * a.o_row_stride;
        k_off = a.block_table ? 0 : (int64_t)k0 * a.k_row_stride;
        v_off = a.block_table ? 0 : (int64_t)k0 * a.v_row_stride;
    } else {
        sq = a.seqlen_q;
        sk = a.seqlens_k ? min(max(a.seqlens_k[b] + a.seqlens_k_offset, 0), a.seqlen_k) : a.seqlen_k;
        q_off = b * a.q_batch_stride;
        o_off = b * a.o_batch_stride;
        k_off = a.block_table ? 0 : b * a.k_batch_stride;
        v_off = a.block_table ? 0 : b * a.v_batch_stride;
    }
    const int nrows = MQ ? a.mq_rows : sq; // rows of the (batch, head) / (batch, KV head) this block is cut from
    if (m0 >= nrows) return; // whole workgroup, before any barrier

    const char* qbase = (const char*)a.q + 2 * (q_off + (int64_t)hq * a.q_head_stride);
    const char* kbase = (const char*)a.k + 2 * (k_off + (int64_t)hk * a.k_head_stride);
    const char* vbase = (const char*)a.v + 2 * (v_off + (int64_t)hk * a.v_head_stride);
    char* obase = (char*)a.o + 2 * (o_off + (int64_t)hq * a.o_head_stride);
    const int32_t* table = a.block_table ? a.block_table + b * a.table_batch_stride : nullptr;

    // row -> query position (what the masks see) and byte offset from qbase / obase.  Prefill: the row IS the
    // position.  MQ: row = position * G + head-in-group.
    const int G = MQ ? a.group : 1;
    auto row_pos = [&](int row) { return MQ ? row / G : row; };
    auto row_off = [&](int row, int64_t row_stride, int64_t head_stride) -> int64_t {
        if constexpr (!MQ) return (int64_t)row * row_stride;
        // (packed rows address one batch element's Q / O: the launcher keeps that below 2^31 elements, so the
        // arithmetic stays in 32 bits and the strides in one scalar register each)
        const int pos = row / G;
        return (int64_t)(pos * (int)row_stride + (row - pos * G) * (int)head_stride);
    };

    // key window of query position r: [r + lo, r + hi] intersected with [0, sk)  (top-left: off = 0, the reference's
    // alignment, prefill.cuh:416-419; bottom-right: off = sk - sq)
    const int off = a.bottom_right ? sk - sq : 0;
    const bool has_hi = a.has_hi, has_lo = a.has_lo;
    const int hi = off + a.hi_off, lo = off + a.lo_off;
    // key tiles this workgroup visits: [j_lo, j_lo + nt)
    int j_lo = 0, nt;
    {
        const int last_pos = row_pos(min(m0 + BM, nrows) - 1);
        const int hi_key = has_hi ? min(sk - 1, last_pos + hi) : sk - 1;
        const int j_hi = hi_key >= 0 ? hi_key / kBN + 1 : 0;
        if (has_lo) j_lo = min(max(row_pos(m0) + lo, 0) / kBN, j_hi);
        nt = j_hi - j_lo;
        if constexpr (MQ) { // this split's share of the tiles (possibly none: it still writes an empty partial)
            const int per = (nt + a.num_splits - 1) / a.num_splits;
            j_lo += split * per;
            nt = max(min(per, j_hi - j_lo), 0);
        }
    }

    // ---- Q fragments (B operand of S^T = K.Q^T): row m0+32*wave+r, columns 16*ks + 8h .. +7 -------
    const int qrow = m0 + 32 * wave + r;
    frag8 qf[KS];
    {
        const char* qp = qbase + 2 * row_off(min(qrow, nrows - 1), a.q_row_stride, a.q_head_stride) + 16 * h;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = MFA_DEV_NT_Q ? __builtin_nontemporal_load((const frag8*)(qp + 32 * ks)) : *(const frag8*)(qp + 32 * ks);
    }

    // ---- staging: global -> LDS directly (LDS-DMA, global_load_lds_dwordx4), no VGPR round trip --------------
    // One wave-instruction writes 1 KiB of LDS contiguously in lane order = RPI = 1024/RB whole tile rows; lane l
    // sits at row (l / LPR), 16-byte position p = l % LPR of that piece.  The LDS image is swizzled (K: chunk ^
    // swz(row) for the ds_read_b128 fragments; V: 64-byte unit ^ swz(row) for the transposed reads), so the lane at
    // position p FETCHES the global chunk that belongs there (swizzle on the source address; the destination
    // cannot scatter).  Rows past the key length are clamped to the last valid row: their scores are masked to
    // -inf, so P is exactly 0 there and the duplicated (finite) V rows contribute nothing.
    const uint32_t smem_lds = lds_address(smem);
    constexpr int LPR_ = RB / 16;            // lanes (16-byte positions) per LDS row
    constexpr int RPI = 64 / LPR_;           // rows per wave-instruction
    constexpr int NI = kBN / (RPI * NW) > 0 ? kBN / (RPI * NW) : 1; // instructions per wave per tile (K, and V)
    static_assert(kBN % (RPI * NW) == 0 || kBN / (RPI * NW) == 0, "tile rows must split evenly over the waves");
    // STREAM (launcher's choice for packed-row launches with ONE row block, where every K/V byte is read exactly
    // once): the DMA uses the non-temporal policy, which lifts the achievable HBM rate from 5.8-6.0 to 6.8-6.9 TB/s
    // (tools/probes/stream_probe.hip).  Prefill, and packed launches whose row blocks re-read K/V out of L2, keep the
    // default policy (prefill with nt: -20...-30 %).
    constexpr bool DMA_NT = STREAM || MFA_DEV_ABL(2048);
    const int last_key = max(sk - 1, 0);
    const int srow = wave * RPI + lane / LPR_; // row of this lane inside instruction 0 (+ i*RPI*NW for instruction i)
    const int spos = lane % LPR_;
    // source chunk for this lane's position; positions whose chunk lies in the row padding fetch chunk 0 (never read).
    // v_swz is invariant under row += RPI*NW for every RB; k_swz is too unless RPI*NW = 8 (RB = 512), where
    // instruction i flips bit 3 of the chunk for odd i.
    constexpr bool KSWZ_FIXED = (RPI * NW) % 16 == 0;
    auto k_src_chunk = [&](int row) { const int ch = spos ^ k_swz<RB>(row); return ch < CH ? ch : 0; };
    const int s_kch = k_src_chunk(srow);
    const int s_vch0 = (((spos >> 2) ^ v_swz<RB>(srow)) << 2) | (spos & 3);
    const int s_vch = s_vch0 < CH ? s_vch0 : 0;
    const uint32_t k_sb = (uint32_t)(2 * a.k_row_stride), v_sb = (uint32_t)(2 * a.v_row_stride); // row pitch, bytes
    const uint32_t k_go = srow * k_sb + 16 * s_kch, v_go = srow * v_sb + 16 * s_vch;
    const uint32_t k_gmax = last_key * k_sb + 16 * s_kch, v_gmax = last_key * v_sb + 16 * s_vch;
    // paged K/V: page id of each of this lane's NI rows of the NEXT tile to be fetched
    // (PG == 1: the page strides live in VGPRs -- these instances sit at the scalar-register limit, where the backend
    // reserves an emergency stack slot it never uses)
    auto in_vgpr = [](int64_t x) {
        uint32_t lo, hi;
        asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=v"(lo), "=v"(hi) : "s"((uint32_t)x), "s"((uint32_t)((uint64_t)x >> 32)));
        return (int64_t)(((uint64_t)hi << 32) | lo);
    };
    const int64_t k_page_bytes = PG == 1 ? in_vgpr(2 * a.k_block_stride) : 0, v_page_bytes = PG == 1 ? in_vgpr(2 * a.v_block_stride) : 0;
    int pid_n[PG == 1 ? NI : 1];
    const char *kpage_n = nullptr, *vpage_n = nullptr; // PG == 2: page bases of the next tile to be fetched
    int tab_chunk = -1, tab_v = 0;                     // PG == 2: 64 block-table entries in a VGPR (see load_pids)
    auto load_pids = [&](int j) {
        if constexpr (PG == 2) { // one page per tile.  The batch element's block-table row sits in ONE VGPR, 64 entries at a
            // time (lane i: entry 64 * chunk + i), and a tile's page id is a v_readlane: a scalar table load per tile is
            // waited for with lgkmcnt(0), which also drains the LDS fragment reads in flight (round 3: paged varlen prefill
            // bf16 B16 S2048 24/8 causal, page 256: 0.50 ms with the per-tile load)
            const int pg = min((j * kBN) >> a.page_shift, a.max_blocks - 1);
            const int chunk = pg >> 6;
            if (chunk != tab_chunk) { // (wave-uniform; once per 64 pages: the vector load's wait also drains DMA in flight)
                tab_chunk = chunk;
                tab_v = table[min(64 * chunk + lane, a.max_blocks - 1)];
            }
            const int64_t pid = __builtin_amdgcn_readlane(tab_v, pg & 63);
            kpage_n = kbase + 2 * pid * a.k_block_stride;
            vpage_n = vbase + 2 * pid * a.v_block_stride;
        } else if constexpr (PG == 1) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int key = min(j * kBN + i * RPI * NW + srow, last_key);
                const int pg = a.page_shift >= 0 ? (key >> a.page_shift) : (key / a.page_size);
                pid_n[i] = table[min(pg, a.max_blocks - 1)];
            }
        }
    };
    // piece pc of tile j: pieces 0..NI-1 are K rows, NI..2*NI-1 are V rows
    auto stage_piece = [&](int j, auto bufc, int pc) {
        constexpr int BUF = decltype(bufc)::value;
        if (NI * RPI * NW > kBN && srow >= kBN) return; // (tiny head dims: fewer rows than lanes cover)
        const bool is_v = pc >= NI;
        const int row = (is_v ? pc - NI : pc) * RPI * NW; // + srow; wave-uniform part
        // wave-uniform LDS piece base, as an integer off the one address conversion done at kernel entry (a
        // generic->LDS pointer cast per piece costs a null check: s_cmp_lg_u64 + s_cselect); readfirstlane makes
        // the uniformity provable for the "s" operand
        const uint32_t dst = __builtin_amdgcn_readfirstlane(
            smem_lds + (is_v ? 2 * TILE_BYTES : 0) + BUF * TILE_BYTES + (row + wave * RPI) * RB);
        if constexpr (PG == 2) {
            // the tile's page id is wave-uniform: page base on the scalar side, the row inside the page as a 32-bit
            // lane offset (the dense path's addressing; rows past the last key clamp to it, in the same page)
            // All of it but one add and one min per piece is scalar: r0 = the piece's first row inside the page, lim = the
            // last key's row there (the page's last row when the last key lies in a later page).  A per-lane row * pitch
            // is a quarter-rate v_mul_lo per piece, which this VALU-co-limited loop does not hide.
            const int key0 = j * kBN + row;
            const uint32_t r0 = (uint32_t)(key0 & (a.page_size - 1));
            const uint32_t lim = (uint32_t)min(last_key - (key0 - (int)r0), a.page_size - 1);
            if (is_v) {
                lds_dma16<DMA_NT>(vpage_n, min(r0 * v_sb + v_go, lim * v_sb + 16 * s_vch), dst);
            } else {
                const int kd = KSWZ_FIXED ? 0 : 16 * (k_src_chunk(srow + row) - s_kch);
                lds_dma16<DMA_NT>(kpage_n, min(r0 * k_sb + k_go, lim * k_sb + 16 * s_kch) + kd, dst);
            }
        } else if constexpr (PG == 1) {
            // page ids were looked up one tile earlier (pid_n): a block-table load right here would be waited for by
            // the compiler with a vmcnt that also drains every DMA piece issued before it
            const int key = min(j * kBN + row + srow, last_key);
            const int pg = a.page_shift >= 0 ? (key >> a.page_shift) : (key / a.page_size);
            const int in = a.page_shift >= 0 ? (key & (a.page_size - 1)) : (key - pg * a.page_size);
            const int64_t pid = pid_n[is_v ? pc - NI : pc];
            if (is_v) lds_dma16<DMA_NT>(vbase + pid * v_page_bytes + ((uint32_t)in * v_sb + 16 * s_vch), dst);
            else lds_dma16<DMA_NT>(kbase + pid * k_page_bytes + ((uint32_t)in * k_sb + 16 * k_src_chunk(srow + row)), dst);
        } else {
            const uint32_t rows = (uint32_t)(j * kBN + row);
            if (is_v) {
                lds_dma16<DMA_NT>(vbase, min(v_go + rows * v_sb, v_gmax), dst);
            } else {
                // (chunk delta of instruction i, a compile-time XOR pattern, when the K swizzle moves with i)
                const int kd = KSWZ_FIXED ? 0 : 16 * (k_src_chunk(srow + row) - s_kch);
                lds_dma16<DMA_NT>(kbase, min(k_go + rows * k_sb, k_gmax) + kd, dst);
            }
        }
    };
    auto stage_dma = [&](int j, auto bufc) {
#pragma unroll
        for (int pc = 0; pc < 2 * NI; ++pc) stage_piece(j, bufc, pc);
    };

    // ---- per-lane LDS read addresses (bases; tile buffer, key block and k-step are immediates) ------
    // K fragment (A operand): row 32*kb + r, chunk (2*ks + h) ^ swz(r)
    const int k_xh = k_swz<RB>(r) ^ h;
    // (absolute LDS pointers, formed once: `smem + offset` inside the loop costs a v_add per access because the
    // dynamic-LDS base is a link-time symbol the compiler cannot fold into the DS immediate)
    const char* k_rd[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) k_rd[ks] = smem + r * RB + 16 * ((2 * ks) ^ k_xh);
    // V^T fragment via transposed read: 16-lane group g16 = lane>>4 : h = g16>>1, column half = g16&1;
    // lane 4q+p of the group addresses key 16*s + 4h + q (+8 for the second read), columns 4p..4p+3 of
    // the 16-column half, i.e. byte (32*db + 16*(g16&1) + 4p)*2 of the row.  The 64-byte unit of column
    // block db is (db ^ swz(key)), i.e. the low two bits of db are XORed: one base per (db & 3).
    const int tq = (lane >> 2) & 3, tp = lane & 3, tcol = (lane >> 4) & 1;
    const int v_key0 = 4 * h + tq;                                 // + 16*s (+8)
    const int v_in64 = (2 * tcol + (tp >> 1)) * 16 + (tp & 1) * 8; // byte inside the 64-byte unit
    const int v_x = v_swz<RB>(v_key0); // depends on key&3 / (key>>1)&1 only: unchanged by +8, +16*s
    constexpr int NVB = DB < 4 ? DB : 4;
    const char* v_rd[NVB];
#pragma unroll
    for (int d = 0; d < NVB; ++d) v_rd[d] = smem + 2 * TILE_BYTES + v_key0 * RB + v_in64 + ((d ^ v_x) * 64);

    float m_run = -INFINITY; // running max of RAW scores (prefill.cuh:454-462)
    float l_run = 0.f;       // this lane's partial row sum (its 32 keys of every tile)
    f32x16 oacc[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[d][i] = 0.f;

    const float c = a.scale_log2;
    const int wrow0 = m0 + 32 * wave; // first query row of this wave
    // query positions of this lane's row and of the wave's first / last row (prefill: the rows themselves)
    const int qpos = row_pos(min(qrow, nrows - 1));
    const int wpos_lo = row_pos(min(wrow0, nrows - 1)), wpos_hi = row_pos(min(wrow0 + 31, nrows - 1));
    const bool wave_has_rows = wrow0 < nrows;

    load_pids(j_lo);
    if (nt > 0 && sk > 0) stage_dma(j_lo, std::integral_constant<int, 0>{});
    load_pids(j_lo + 1);
    // Everything issued so far (Q fragments, tile 0's DMA) is waited for HERE: with the Q loads still on the
    // scoreboard at the loop header, hipcc re-waits for them inside the loop (vmcnt(7..0) before the QK^T
    // MFMAs), which from the second iteration on would drain the just-issued DMA of the next tile.
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
    __syncthreads();
    MFA_DEV_STAMP(1);

    // one key tile out of LDS buffer BUF (compile-time, so every LDS offset is an immediate)
    auto tile = [&](int jj, auto bufc) {
        constexpr int BUF = decltype(bufc)::value;
        const int j = j_lo + jj; // key tile index; jj counts from this block's first tile (buffer parity)
        const bool more = jj + 1 < nt;
        // The next tile's DMA goes into the other buffer (every wave left it at the previous barrier).  Its
        // pieces are issued one per QK^T k-step, between the MFMAs, rather than as a burst: a burst of 1-KiB pieces
        // stalls the wave at issue (the price of a piece depends on what else is in flight).
        constexpr auto nbuf = std::integral_constant<int, BUF ^ 1>{};
        const bool dma = more && !MFA_DEV_ABL(1);

        // a wave none of whose rows can see this tile (above the causal diagonal / outside the window) skips it
        const bool active = (!MQ || wave_has_rows) && (!has_hi || j * kBN <= wpos_hi + hi) &&
                            (!has_lo || j * kBN + kBN - 1 >= wpos_lo + lo);
        if (!active && dma) stage_dma(j + 1, nbuf);
        if (!active) load_pids(j + 2);
        if (active) {
            constexpr int kt = BUF * TILE_BYTES; // byte offset of this tile's K (and, past sK, V) buffer
            constexpr int vt = BUF * TILE_BYTES;
            f32x16 s[2];
#pragma unroll
            for (int i = 0; i < 16; ++i) s[0][i] = s[1][i] = 0.f;
            // K fragments are read PF k-steps ahead of the MFMAs that use them
            constexpr int PF = KS < 2 ? KS : 2;
            frag8 kf[KS][2];
#pragma unroll
            for (int ks = 0; ks < PF; ++ks)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) kf[ks][kb] = *(const frag8*)(k_rd[ks] + kt + kb * 32 * RB);
            // (two copies of the k-step loop, with and without the DMA pieces: a per-step `if (dma)` would cut the
            // MFMA / ds_read stream into KS basic blocks)
            auto qk_steps = [&](auto with_dma) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    if (ks + PF < KS) {
#pragma unroll
                        for (int kb = 0; kb < 2; ++kb)
                            kf[ks + PF][kb] = *(const frag8*)(k_rd[ks + PF] + kt + kb * 32 * RB);
                    }
                    s[0] = E::mfma32(kf[ks][0], qf[ks], s[0]);
                    if constexpr (decltype(with_dma)::value) {
#pragma unroll
                        for (int pc = ks * 2 * NI / KS; pc < (ks + 1) * 2 * NI / KS; ++pc) stage_piece(j + 1, nbuf, pc);
                    }
                    s[1] = E::mfma32(kf[ks][1], qf[ks], s[1]);
                }
            };
            if (dma) qk_steps(std::true_type{});
            else qk_steps(std::false_type{});
            load_pids(j + 2); // (paged) consumed by the next tile's DMA, after this tile's end-of-tile wait
            // mask: key > row + hi (causal: hi = 0, top-left) or key >= sk; register i of block kb is key
            // j*64 + 4h + (32*kb + (i&3) + 8*(i>>2))
            const bool need_mask = (has_hi && j * kBN + kBN - 1 > wpos_lo + hi) || (j + 1) * kBN > sk;
            if (need_mask) {
                const int lim = (has_hi ? min(qpos + hi, sk - 1) : sk - 1) - j * kBN - 4 * h; // keys <= lim stay
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        if (32 * kb + (i & 3) + 8 * (i >> 2) > lim) s[kb][i] = -INFINITY;
            }
            // sliding window: key < row + lo
            if (has_lo && j * kBN < wpos_hi + lo) {
                const int liml = qpos + lo - j * kBN - 4 * h; // keys >= liml stay
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        if (32 * kb + (i & 3) + 8 * (i >> 2) < liml) s[kb][i] = -INFINITY;
            }
            // ---- online softmax, all in this lane's registers ---------------------------------------
            float mt = s[0][0];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) mt = fmaxf(mt, s[kb][i]);
            mt = fmaxf(mt, swap32(mt));
            const float m_new = fmaxf(m_run, mt);
            // Deferred rescale: O and l are rescaled (and the reference max moved) only when some row of this wave
            // grew its max by more than THR in the exponent; otherwise the old max stays and P may reach 2^THR
            // instead of 1 -- the same relative precision in fp16/bf16 and in the fp32 sums, one O-wide multiply
            // pass less per tile.  THR = 0 is the textbook update (every growth rescales).
            constexpr float THR = MFA_DEV_ABL(512) ? 0.f : 6.f;
            if (__builtin_amdgcn_ballot_w64((m_new - m_run) * c > THR) != 0) {
                const float msn = (m_new == -INFINITY) ? 0.f : m_new;
                const float alpha = fast_exp2((m_run - msn) * c);
                l_run *= alpha;
#pragma unroll
                for (int d = 0; d < DB; ++d)
#pragma unroll
                    for (int i = 0; i < 16; ++i) oacc[d][i] *= alpha;
                m_run = m_new;
            }
            const float ms = (m_run == -INFINITY) ? 0.f : m_run;
            const float mc = ms * c;
            // p = exp2(s*c - m*c): packed fma / packed add on register pairs, one v_exp per element
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const f32x2 c2 = {c, c}, nmc2 = {-mc, -mc};
            f32x2 ps2 = {0.f, 0.f};
            uint32_t pk[2][8];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const f32x2 sv = {s[kb][2 * i], s[kb][2 * i + 1]};
                    const f32x2 t = __builtin_elementwise_fma(sv, c2, nmc2);
                    const f32x2 pv = MFA_DEV_ABL(4) ? t : f32x2{fast_exp2(t[0]), fast_exp2(t[1])};
                    ps2 += pv;
                    pk[kb][i] = E::pack(pv[0], pv[1]);
                }
            l_run += ps2[0] + ps2[1];

            // ---- O^T += V^T . P^T : 4 k-steps of 16 keys x DB column blocks ---------------------------
#pragma unroll
            for (int s16 = 0; s16 < 4; ++s16) {
                const int kb = s16 >> 1, sh = s16 & 1;
                u32x4 pw = {pk[kb][4 * sh], pk[kb][4 * sh + 1], pk[kb][4 * sh + 2], pk[kb][4 * sh + 3]};
                const frag8 pf = __builtin_bit_cast(frag8, pw);
#pragma unroll
                for (int d = 0; d < DB; ++d) {
                    const char* va = v_rd[d & 3] + vt + (d >> 2) * 256 + s16 * 16 * RB;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(va));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(va + 8 * RB));
                    typedef short s16x8 __attribute__((ext_vector_type(8)));
                    const s16x8 vv = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    if (!MFA_DEV_ABL(8)) oacc[d] = E::mfma32(__builtin_bit_cast(frag8, vv), pf, oacc[d]);
                    else oacc[d][0] += (float)vv[0] + (float)vv[4];
                }
            }
        }
        // the DMA is a pending LDS write on the VM counter: drain it, then let the other waves read the tile
        if (!MFA_DEV_ABL(64)) __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
        if (!MFA_DEV_ABL(2)) __syncthreads();
    };

    if (MFA_DEV_ABL(256)) nt = 0;
    for (int j = 0; j < nt; j += 2) {
        tile(j, std::integral_constant<int, 0>{});
        if (j + 1 < nt) tile(j + 1, std::integral_constant<int, 1>{});
    }

    MFA_DEV_STAMP(2);
    // ---- epilogue: 1/l (prefill.cuh:600-612), O^T -> LDS rows -> coalesced 16-byte stores ------------
    const float l_tot = l_run + swap32(l_run);
    const float inv = (l_tot == 0.f || l_tot != l_tot) ? 1.f : 1.f / l_tot;
    const float lse_row = l_tot > 0.f ? m_run * a.scale + __logf(l_tot) : -INFINITY; // natural log, -inf: no key seen
    if constexpr (MQ) {
        const int grp = qrow - qpos * G; // head within the group (rows past the block: unused)
        if (a.num_splits > 1) {
            // normalised partial of this key split, fp32, straight from the accumulators: (split, B, Sq, H) rows
            if (qrow < nrows) {
                const int64_t arow = (((int64_t)split * a.batch + b) * sq + qpos) * a.heads + hq + grp;
                if (h == 0) a.lse_acc[arow] = lse_row;
                float* op = a.o_acc + arow * D + 4 * h;
#pragma unroll
                for (int d = 0; d < DB; ++d)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        f32x4 w = {oacc[d][4 * g4] * inv, oacc[d][4 * g4 + 1] * inv, oacc[d][4 * g4 + 2] * inv, oacc[d][4 * g4 + 3] * inv};
                        *(f32x4*)(op + 32 * d + 8 * g4) = w;
                    }
            }
            if (!a.split_ctr) return; // the caller merges the partials (decode_combine_kernel)
            // The last key split of this row block to arrive merges them here: partials released (agent scope), one
            // arrival ticket per workgroup, the winner acquires, resets the counter for the next launch and runs one wave
            // per row over the block's rows.  (All splits of a (batch, KV head) run on one XCD: the partials sit in its L2.)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's partial stores are in the XCD's L2
            __syncthreads();
            int* const flag = (int*)smem; // (the K/V buffers are free: the loop's last barrier)
            const int ctr_idx = (b * a.kv_heads + hk) * a.mq_row_blocks + m0 / BM;
            if (tid == 0) *flag = atomicAdd(a.split_ctr + ctr_idx, 1);
            __syncthreads();
            if (*flag != a.num_splits - 1) return;
            // (no acquire fence: the winner reads the partials with L2-served loads, combine_row<T, true>)
            if (tid == 0) a.split_ctr[ctr_idx] = 0;
            const int64_t BH = (int64_t)a.batch * sq * a.heads;
            for (int row = m0 + wave; row < min(m0 + BM, nrows); row += NW) {
                const int pos = row / G, hh = hq + (row - pos * G);
                const int64_t bh = ((int64_t)b * sq + pos) * a.heads + hh;
                char* orow = obase + 2 * ((int64_t)pos * a.o_row_stride + (int64_t)(row - pos * G) * a.o_head_stride);
                float* lse_out = a.lse ? a.lse + ((int64_t)b * a.heads + hh) * sq + pos : nullptr;
                combine_row<T, true>(a.o_acc, a.lse_acc, a.num_splits, BH, bh, D, orow, lse_out, lane);
            }
            return;
        }
        if (a.lse && h == 0 && qrow < nrows) a.lse[((int64_t)b * a.heads + hq + grp) * sq + qpos] = lse_row;
    } else if (!MFA_DEV_TIMELINE_ON && a.lse && h == 0 && qrow < sq) {
        const int64_t idx = a.cu_q ? (int64_t)hq * a.total_q + a.cu_q[b] + qrow
                                   : ((int64_t)b * a.heads + hq) * a.seqlen_q + qrow;
        a.lse[idx] = lse_row;
    }
    // the loop's last barrier guarantees every wave is done with the K/V buffers
    char* so = smem + wave * 32 * RB; // this wave's 32 rows x RB bytes
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            // registers 4*g4..4*g4+3 are columns 32*d + 8*g4 + 4*h + 0..3 of row r
            u32x2 w;
            w[0] = E::pack(oacc[d][4 * g4] * inv, oacc[d][4 * g4 + 1] * inv);
            w[1] = E::pack(oacc[d][4 * g4 + 2] * inv, oacc[d][4 * g4 + 3] * inv);
            const int ch = 4 * d + g4;
            *(u32x2*)(so + r * RB + 16 * (ch ^ k_swz<RB>(r)) + 8 * h) = w;
        }
    // same wave wrote and reads: LDS ops of one wave complete in order
    constexpr int ROWS_PER_IT = 64 / CH > 0 ? 64 / CH : 1;
    if constexpr (64 % CH == 0) {
        const int rr0 = lane / CH, ch = lane % CH;
#pragma unroll
        for (int it = 0; it < 32 / ROWS_PER_IT; ++it) {
            const int rr = it * ROWS_PER_IT + rr0;
            const u32x4 val = *(const u32x4*)(so + rr * RB + 16 * (ch ^ k_swz<RB>(rr)));
            const int grow = wrow0 + rr;
            if (grow < nrows) {
                u32x4* dst = (u32x4*)(obase + 2 * row_off(grow, a.o_row_stride, a.o_head_stride) + 16 * ch);
                if (MFA_DEV_NT_O) __builtin_nontemporal_store(val, dst);
                else *dst = val;
            }
        }
    } else {
        // CH does not divide 64 (D = 96, 160, ...): walk the 32*CH chunks of the wave linearly
        for (int idx = lane; idx < 32 * CH; idx += 64) {
            const int rr = idx / CH, ch = idx - rr * CH;
            const u32x4 val = *(const u32x4*)(so + rr * RB + 16 * (ch ^ k_swz<RB>(rr)));
            const int grow = wrow0 + rr;
            if (grow < nrows) {
                u32x4* dst = (u32x4*)(obase + 2 * row_off(grow, a.o_row_stride, a.o_head_stride) + 16 * ch);
                if (MFA_DEV_NT_O) __builtin_nontemporal_store(val, dst);
                else *dst = val;
            }
        }
    }
    MFA_DEV_STAMP_FLUSH(a.lse, tid, nt);
}

template <typename T, int D, int NW, int PG>
static int launch_prefill_p(PrefillArgs& a, hipStream_t stream) {
    constexpr int BM = 32 * NW;
    constexpr int RB = Pitch<D>::RB;
    constexpr size_t smem = 4 * kBN * RB;
    a.num_m_blocks = (a.seqlen_q + BM - 1) / BM;
    const int64_t npairs = (int64_t)a.heads * a.batch;
    if (npairs <= 0 || a.num_m_blocks <= 0) return 0;
    const int knob_gp = g_knobs.group_pairs.load();
    a.group_pairs = knob_gp > 0 ? knob_gp : 4;
    // every XCD gets ceil(npairs / 8) pairs' worth of slots, rounded up to whole groups; surplus blocks exit
    // (lengths differ by batch element: pairs dealt round-robin over the XCDs, see the kernel)
    // fewer than 8 pairs (one long prompt on a tensor-parallel shard's few heads): an XCD per pair would leave XCDs idle; a small
    // pair count that is no multiple of 8 would load them unevenly (12 pairs: four XCDs with two, four with one).  The row
    // blocks of all pairs go out in plain order instead (3)
    a.interleave_pairs = npairs < 64 && (npairs & 7) ? 3 : !(a.cu_q || a.seqlens_k) ? 0 : (int64_t)a.batch * a.kv_heads >= 16 ? 2 : 1;
    const int64_t per_xcd = a.interleave_pairs == 2 ? (((int64_t)a.batch * a.kv_heads + 7) / 8) * a.group : (npairs + 7) / 8;
    const int64_t groups = (per_xcd + a.group_pairs - 1) / a.group_pairs;
    const int64_t total = a.interleave_pairs == 3 ? npairs * a.num_m_blocks : 8 * groups * a.group_pairs * a.num_m_blocks;
    if (total > 0x7fffffffLL) return -1;
    auto kern = prefill_fwd_kernel<T, D, NW, PG>;
    if (smem > 64 * 1024 &&
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
        return -3;
    hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(64 * NW), smem, stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

template <typename T, int D, int NW>
static int launch_prefill_t(PrefillArgs& a, hipStream_t stream) {
    if (!a.block_table) return launch_prefill_p<T, D, NW, 0>(a, stream);
    return a.page_shift >= 6 ? launch_prefill_p<T, D, NW, 2>(a, stream) : launch_prefill_p<T, D, NW, 1>(a, stream);
}

template <typename T>
static int launch_prefill_d(PrefillArgs& a, hipStream_t stream) {
    switch (a.head_dim) {
    case 32: return launch_prefill_t<T, 32, 4>(a, stream);
    case 64: return launch_prefill_t<T, 64, 4>(a, stream);
    case 96: return launch_prefill_t<T, 96, 4>(a, stream);
    case 128: {
        return g_knobs.nw8.load() ? launch_prefill_t<T, 128, 8>(a, stream) : launch_prefill_t<T, 128, 4>(a, stream);
    }
    case 160: return launch_prefill_t<T, 160, 4>(a, stream);
    case 192: return launch_prefill_t<T, 192, 4>(a, stream);
    case 224: return launch_prefill_t<T, 224, 4>(a, stream);
    case 256: return launch_prefill_t<T, 256, 4>(a, stream);
    default: return -2;
    }
}

static PrefillArgs make_args(const mfa_forward_params& p) {
    PrefillArgs a{};
    a.q = p.q_ptr; a.k = p.k_ptr; a.v = p.v_ptr; a.o = p.o_ptr;
    a.cu_q = p.cu_seqlens_q; a.cu_k = p.cu_seqlens_k; a.block_table = p.block_table;
    a.q_batch_stride = p.q_batch_stride; a.q_head_stride = p.q_head_stride; a.q_row_stride = p.q_row_stride;
    a.k_batch_stride = p.k_batch_stride; a.k_head_stride = p.k_head_stride; a.k_row_stride = p.k_row_stride;
    a.v_batch_stride = p.v_batch_stride; a.v_head_stride = p.v_head_stride; a.v_row_stride = p.v_row_stride;
    a.o_batch_stride = p.o_batch_stride; a.o_head_stride = p.o_head_stride; a.o_row_stride = p.o_row_stride;
    a.k_block_stride = p.k_cache_block_stride; a.v_block_stride = p.v_cache_block_stride;
    a.table_batch_stride = p.block_table_batch_stride;
    a.batch = p.batch; a.heads = p.heads; a.kv_heads = p.kv_heads; a.group = p.heads / p.kv_heads;
    a.head_dim = p.head_dim; a.seqlen_q = p.seqlen_q; a.seqlen_k = p.seqlen_k;
    a.page_size = p.page_block_size > 0 ? p.page_block_size : 1;
    a.page_shift = (a.page_size & (a.page_size - 1)) == 0 ? __builtin_ctz(a.page_size) : -1;
    a.max_blocks = p.max_blocks_per_seq > 0 ? p.max_blocks_per_seq : 0x7fffffff;
    a.is_causal = p.is_causal;
    a.scale_log2 = p.softmax_scale_log2;
    a.scale = p.softmax_scale;
    a.seqlens_k = p.cu_seqlens_k ? nullptr : p.seqlens_k;
    a.seqlens_k_offset = p.seqlens_k_offset;
    a.lse = p.softmax_lse_ptr;
    a.total_q = p.total_q;
    a.bottom_right = p.mask_bottom_right;
    // effective key window: causal closes the right side at the diagonal; the opt-in local window (not the
    // reference's window_size_* fields, which it accepts and ignores) narrows either side
    int wl = -1, wr = p.is_causal ? 0 : -1;
    if (p.use_local_window) {
        wl = p.local_window_left;
        if (p.local_window_right >= 0) wr = wr < 0 ? p.local_window_right : (p.local_window_right < wr ? p.local_window_right : wr);
    }
    a.has_hi = wr >= 0; a.hi_off = wr >= 0 ? wr : 0;
    a.has_lo = wl >= 0; a.lo_off = wl >= 0 ? -wl : 0;
    return a;
}

int launch_prefill(const mfa_forward_params& p, hipStream_t stream, bool* used_prefill64) {
    PrefillArgs a = make_args(p);
    // head dim 128, dense: the 64-rows-per-wave kernel (mfa_prefill64.hip) from 512 keys up with a right bound (causal), from
    // 384 without.  Below that its longer way in and out of a work item costs more than its loop gains (fp16 B48 H24, same box,
    // tools/short_s.py, general / 64-row kernel, us: S=128 27 / 43, S=256 causal 51 / 67, non-causal 59 / 68, S=320 88 / 102
    // and 105 / 107, S=384 106 / 112 and 131 / 122, S=512 153 / 147 and 209 / 187).  MFA_PREFILL64=0 forces the general
    // kernel, =2 the 64-row one for everything it serves.
    static const int env_p64 = [] { const char* e = getenv("MFA_PREFILL64"); return e ? atoi(e) : 1; }();
    // Packed variable-length batches take it too (round 3) when they are even enough for its static schedule.  The lengths
    // are device data, so the launcher goes by the mean query length total_q / batch: at least 0.9 of max_seqlen_q.  Ragged
    // batches keep the general kernel, whose workgroups the hardware deals out as CUs fall free (bf16 H24/8 causal,
    // tools/varlen_point.py, general / 64-row kernel ms: 16 x 2048 0.472 / 0.417, 48 x 1024 0.421 / 0.387; lengths
    // 1024..2048 0.477 / 0.505, 512..4096 0.774 / 0.902, 128..4096 log-uniform 1.21 / 1.61).
    const int p64_from = a.has_hi ? 512 : 384;
    bool p64_fits = a.seqlen_k >= p64_from;
    if (a.cu_q) {
        const bool even = a.batch > 0 && 10 * a.total_q >= 9 * (int64_t)a.batch * a.seqlen_q;
        a.p64_ragged = !even;
        // ragged: the schedule built from the lengths (at most 256 sequences of at most 64 row blocks) where long sequences carry the
        // work -- a mean length of 512, or a longest sequence that outweighs the rest even if the rest were all 256 long
        // (the general kernel is the faster one on sequences that short)
        const int64_t rest = a.total_q - a.seqlen_q;
        const bool long_heavy = a.total_q >= 512 * (int64_t)a.batch || (int64_t)a.seqlen_q * a.seqlen_q >= 256 * rest;
        p64_fits = p64_fits && a.seqlen_q >= p64_from && (even || (a.batch <= 256 && a.seqlen_q <= 64 * 256 && long_heavy));
    }
    if (env_p64 == 2 || (env_p64 == 1 && p64_fits)) {
        a.p64_forced = env_p64 == 2;
        const int rc = launch_prefill64(a, p.is_bf16 != 0, stream);
        if (rc != -2) {
            if (used_prefill64) *used_prefill64 = rc == 0;
            return rc;
        }
    }
    return p.is_bf16 ? launch_prefill_d<BFloat>(a, stream) : launch_prefill_d<Half>(a, stream);
}

// ---- packed-row kv-cache attention ---------------------------------------------------------------------------
template <typename T, int D, int PG>
static int launch_mq_p(PrefillArgs& a, hipStream_t stream) {
    constexpr int NW = 4, BM = 32 * NW;
    constexpr int RB = Pitch<D>::RB;
    constexpr size_t smem = 4 * kBN * RB;
    a.mq_rows = a.seqlen_q * a.group;
    if ((int64_t)a.seqlen_q * std::max(a.q_row_stride, a.o_row_stride) + (int64_t)a.group * std::max(a.q_head_stride, a.o_head_stride) >= (1LL << 31)) return -2;
    a.mq_row_blocks = (a.mq_rows + BM - 1) / BM;
    const int64_t npairs = (int64_t)a.batch * a.kv_heads;
    if (npairs <= 0 || a.mq_rows <= 0) return 0;
    const int64_t total = (spread_splits(npairs, a.num_splits) ? npairs : 8 * ((npairs + 7) / 8)) * a.num_splits * a.mq_row_blocks;
    if (total > 0x7fffffffLL) return -1;
    auto kern = prefill_fwd_kernel<T, D, NW, PG, true, false>;
    const int env_nt = g_knobs.mq_stream.load();
    if (env_nt == 1 || (env_nt != 0 && a.mq_row_blocks == 1)) kern = prefill_fwd_kernel<T, D, NW, PG, true, true>;
    if (smem > 64 * 1024 &&
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
        return -3;
    hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(64 * NW), smem, stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

template <typename T>
static int launch_mq_d(PrefillArgs& a, hipStream_t stream) {
    const int pg = !a.block_table ? 0 : (a.page_shift >= 6 ? 2 : 1); // (page_shift >= 0: page_size = 2^page_shift)
#define MFA_MQ_CASE(DD)                                                                                                 \
    case DD:                                                                                                          \
        return pg == 0 ? launch_mq_p<T, DD, 0>(a, stream) : pg == 1 ? launch_mq_p<T, DD, 1>(a, stream) : launch_mq_p<T, DD, 2>(a, stream)
    switch (a.head_dim) { // (160, 192, 224 take the per-head prefill path: the caller's fallback)
        MFA_MQ_CASE(32);
        MFA_MQ_CASE(64);
        MFA_MQ_CASE(96);
        MFA_MQ_CASE(128);
        MFA_MQ_CASE(256);
    default: return -2;
    }
#undef MFA_MQ_CASE
}

// The in-kernel split merge orders partial stores and the arrival counter through ONE L2: it needs every key split of a
// (batch, KV head) on the same XCD, which the launch arranges by workgroup id (id & 7 = XCD, the dispatcher's round robin).
// That holds on every MI355X partition mode seen, but correctness must not hang on it: mfa_init() (include/mfa.h) checks it
// per device with the hardware's XCC_ID register; until it has, and where it fails, the merge stays a separate launch.
// No launch entry point runs the probe: it allocates and synchronises.
__global__ void xcd_probe_kernel(int* mismatches) {
    const int xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf; // HW_REG_XCC_ID
    __shared__ int first;
    // (which XCD workgroup 0 lands on is not fixed: compare against the XCD of the workgroup's own residue class, recorded
    //  by its first member to arrive)
    if (threadIdx.x == 0) {
        int* slot = mismatches + 1 + (blockIdx.x & 7);
        const int seen = atomicCAS(slot, -1, xcc);
        first = seen == -1 ? xcc : seen;
        if (first != xcc) atomicAdd(mismatches, 1);
    }
}
namespace {
constexpr int kMaxDevices = 64;
std::atomic<int> g_xcd_state[kMaxDevices]; // 0 = not probed, 1 = premise holds, 2 = it does not
}
int xcd_premise_probe(int dev) {
    if (dev < 0 || dev >= kMaxDevices) return 0;
    const int st = g_xcd_state[dev].load();
    if (st) return st == 1;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    if (g_xcd_state[dev].load()) return g_xcd_state[dev].load() == 1;
    int prev = 0;
    if (hipGetDevice(&prev) != hipSuccess || hipSetDevice(dev) != hipSuccess) return -1;
    int* d = nullptr;
    int h[9];
    int rc = -1;
    if (hipMalloc(&d, sizeof(h)) == hipSuccess) {
        h[0] = 0;
        for (int i = 1; i < 9; ++i) h[i] = -1;
        if (hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice) == hipSuccess) {
            hipLaunchKernelGGL(xcd_probe_kernel, dim3(4096), dim3(64), 0, 0, d);
            if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
                bool distinct = true; // eight residue classes on eight different XCDs, no workgroup off its class's XCD
                for (int i = 1; i < 9; ++i)
                    for (int j = i + 1; j < 9; ++j) distinct = distinct && h[i] != h[j];
                rc = h[0] == 0 && distinct ? 1 : 0;
            }
        }
        (void)hipFree(d);
    }
    (void)hipSetDevice(prev);
    if (rc >= 0) g_xcd_state[dev].store(rc == 1 ? 1 : 2);
    return rc;
}

// Whether the merge of `splits` key splits runs inside the split kernel for a launch of `workgroups` workgroups writing
// `partial_bytes` of fp32 partials, and with which counters: the caller's (mfa_forward_params::split_counters, zeroed, at
// least n entries), provided mfa_init() found the XCD premise to hold on this device.  Otherwise null: the caller of this
// function launches decode_combine_kernel behind the split kernel.
bool fused_merge_pays(int64_t units, int64_t workgroups, int64_t pbytes) {
    // Measured (tools/ab_decode_map.py, one box, interleaved rounds; profiles/r03a_ab_decode_map_and_merge.txt): the
    // in-kernel merge saves 1.0-1.3 us of 15-30 us on launches of at most one round of workgroups (256: B4 Skv8192, B8
    // Skv4096, B16 Skv2048; 512 on the packed kernel: BASELINE config 5, 50.9 -> 49.2 us) and LOSES where later rounds
    // of streaming workgroups queue behind the winners' invalidate and re-read: 768 workgroups +1.0 us (config 3 forced
    // to 4 splits), 1 152 +6.6 us of 32 (README MHA B24 H24 Skv512), 1 536 +0.9 us (G=8, Skv8192).
    return !spread_splits(units, 2) && workgroups <= kFusedMergeMaxWorkgroups && pbytes <= kFusedMergeMaxPartialBytes;
}
int32_t* pick_split_counters(const mfa_forward_params& p, size_t n, int64_t units, int64_t workgroups, int64_t pbytes) {
#ifdef MFA_DEV_DECODE_AB // developer A/B builds: read per launch; =2: no size gate
    const int env = [] { const char* e = getenv("MFA_FUSED_COMBINE"); return e ? atoi(e) : 1; }();
#else
    static const int env = [] { const char* e = getenv("MFA_FUSED_COMBINE"); return e ? atoi(e) : 1; }();
#endif
    if (!env || !p.split_counters || (size_t)(p.split_counters_len < 0 ? 0 : p.split_counters_len) < n) return nullptr;
    if (spread_splits(units, 2) || (env != 2 && !fused_merge_pays(units, workgroups, pbytes))) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices || g_xcd_state[dev].load() != 1) return nullptr;
    return p.split_counters;
}

int launch_kvcache_packed(const mfa_forward_params& p, hipStream_t stream, bool* merged_in_kernel) {
    PrefillArgs a = make_args(p);
    a.cu_q = a.cu_k = nullptr;
    a.num_splits = p.num_splits < 1 ? 1 : p.num_splits;
    a.o_acc = p.oaccum_ptr;
    a.lse_acc = p.softmax_lseaccum_ptr;
    a.split_ctr = nullptr;
    if (a.num_splits > 1) {
        const int64_t row_blocks = ((int64_t)a.seqlen_q * a.group + 127) / 128;
        const int64_t n = (int64_t)a.batch * a.kv_heads * row_blocks;
        a.split_ctr = pick_split_counters(p, (size_t)n, (int64_t)a.batch * a.kv_heads, n * a.num_splits, partial_bytes(p));
    }
    const int rc = p.is_bf16 ? launch_mq_d<BFloat>(a, stream) : launch_mq_d<Half>(a, stream);
    if (rc) return rc;
    if (merged_in_kernel) *merged_in_kernel = a.num_splits > 1 && a.split_ctr != nullptr;
    return a.num_splits > 1 && !a.split_ctr ? launch_decode_combine(p, stream) : 0;
}

} // namespace mfa
