/*
 * mfa_oracle.c — CPU restatement of the reference's attention-forward algorithm.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call this
 * file.  The product path (libmfa_hip.so / mini_flash_attention._C) never links or falls back to it.
 *
 * What it restates (reference = w4096/mini-flash-attention, CUDA; SURVEY.md Appendix A is the spec):
 *   prefill  csrc/mfa/prefill.cuh:452-483 (online softmax update on RAW scores, exp2 with scale*log2e),
 *            :393-421 (top-left causal + key-length mask), :549-612 (P rounded to the element type before
 *            P.V, fp32 O, 1/l with the 0/NaN guard), :745-752 (64-key tile range), scales api.cpp:99-100
 *   decode   csrc/mfa/decode.cuh:296-312 (fp32 dot), :367-383 (per-warp online softmax, rows k = 4n + w,
 *            P kept fp32), :587-661 (4-warp merge, lse = M*scale + ln L), :26-30 (split tile ranges),
 *            :718-747 (combine; max-subtracted here, the reference is not)
 *   paging   key t of sequence b lives at block_table[b][t / page] * page + t % page, resolved per KEY
 *            (the reference resolves per 64-key tile, which is only right when page % 64 == 0)
 * Deliberate deviations from reference quirks: the running max starts at -inf
 * (reference: FLT_MIN) with an explicit fully-masked-row guard; the causal tile range is clamped to the
 * key length; cache_seqlens == NULL means "whole cache".
 *
 * PARITY PIN: the reference cannot be compiled here (needs nvcc + the un-vendored CUTLASS submodule,
 * SURVEY.md §8c) nor imported (its Python layer needs the compiled _C), and it ships no golden vectors.
 * This restatement is pinned against the oracle the reference's own tests use — torch SDPA in fp32
 * (tests/test_mha.py:75-91, test_causal.py:87-93, test_gqa.py:118-128) — at the reference's test shapes
 * and thresholds (tests/test_oracle_cpu.py).  For varlen / paged / decode values the reference's tests
 * pin only through the absent `flash_attn` package: at the reference boundary those paths are
 * "parity unpinned"; they are pinned to SDPA here instead.
 *
 * Build: gcc -O2 -fPIC -shared -fopenmp (optional) oracle/mfa_oracle.c -o oracle/_build/libmfa_oracle.so -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/mfa.h"

#define TILE 64

/* ---- 16-bit float conversions (round-to-nearest-even) ---------------------------------------- */
static float bits_f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static uint32_t f_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static float bf16_to_f(uint16_t h) { return bits_f((uint32_t)h << 16); }
static uint16_t f_to_bf16(float f) {
    uint32_t u = f_bits(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40); /* quiet NaN */
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float f16_to_f(uint16_t h) {
    const uint32_t s = (uint32_t)(h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    if (e == 0) {
        if (m == 0) return bits_f(s);
        e = 1;
        while (!(m & 0x400u)) { m <<= 1; --e; }
        m &= 0x3ffu;
        return bits_f(s | ((e + 112u) << 23) | (m << 13));
    }
    if (e == 31) return bits_f(s | 0x7f800000u | (m << 13));
    return bits_f(s | ((e + 112u) << 23) | (m << 13));
}
static uint16_t f_to_f16(float f) {
    const uint32_t u = f_bits(f);
    const uint16_t s = (uint16_t)((u >> 16) & 0x8000u);
    const uint32_t a = u & 0x7fffffffu;
    if (a > 0x7f800000u) return (uint16_t)(s | 0x7e00u);
    if (a >= 0x47800000u) return (uint16_t)(s | 0x7c00u); /* >= 65536 -> inf (65520 rounds to inf below) */
    if (a < 0x33000001u) return s;                        /* < 2^-25 (or exactly 2^-25, ties to even 0) */
    int e = (int)(a >> 23) - 127;
    uint32_t m = (a & 0x7fffffu) | 0x800000u;
    int shift = (e < -14) ? (13 + (-14 - e)) : 13;
    uint32_t half = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1u), mid = 1u << (shift - 1);
    if (rem > mid || (rem == mid && (half & 1u))) ++half;
    if (e < -14) return (uint16_t)(s | half); /* subnormal (may carry into the smallest normal) */
    uint32_t r = ((uint32_t)(e + 15) << 10) + (half - 0x400u);
    if (r >= 0x7c00u) r = 0x7c00u;
    return (uint16_t)(s | r);
}
static float ld(const void* base, int64_t idx, int bf16) {
    const uint16_t h = ((const uint16_t*)base)[idx];
    return bf16 ? bf16_to_f(h) : f16_to_f(h);
}
static void st(void* base, int64_t idx, float v, int bf16) {
    ((uint16_t*)base)[idx] = bf16 ? f_to_bf16(v) : f_to_f16(v);
}
static float round_elt(float v, int bf16) { return bf16 ? bf16_to_f(f_to_bf16(v)) : f16_to_f(f_to_f16(v)); }

/* exported for the conversion unit tests */
uint16_t mfa_oracle_f32_to_f16(float f) { return f_to_f16(f); }
uint16_t mfa_oracle_f32_to_bf16(float f) { return f_to_bf16(f); }
float mfa_oracle_f16_to_f32(uint16_t h) { return f16_to_f(h); }
float mfa_oracle_bf16_to_f32(uint16_t h) { return bf16_to_f(h); }

/* element offset of key row t of sequence b (dense, varlen-packed or paged) */
static int64_t kv_row_offset(const mfa_forward_params* p, int b, int t, int64_t batch_stride, int64_t row_stride,
                             int64_t block_stride) {
    if (p->block_table) {
        const int pg = t / p->page_block_size, in = t % p->page_block_size;
        const int64_t pid = p->block_table[(int64_t)b * p->block_table_batch_stride + pg];
        return pid * block_stride + (int64_t)in * row_stride;
    }
    if (p->cu_seqlens_k) return ((int64_t)p->cu_seqlens_k[b] + t) * row_stride;
    return (int64_t)b * batch_stride + (int64_t)t * row_stride;
}

/* ---- prefill / varlen / paged prefill -------------------------------------------------------- */
static void prefill_head(const mfa_forward_params* p, int b, int h, int sq, int sk, int64_t qo, int64_t oo) {
    const int D = p->head_dim, bf = p->is_bf16, G = p->heads / p->kv_heads, hk = h / G;
    const float c = p->softmax_scale_log2;
    float* vbuf = (float*)malloc(sizeof(float) * (size_t)TILE * D);
    float* acc = (float*)malloc(sizeof(float) * D);
    float* qrow = (float*)malloc(sizeof(float) * D);
    for (int r = 0; r < sq; ++r) {
        for (int d = 0; d < D; ++d) {
            qrow[d] = ld(p->q_ptr, qo + (int64_t)r * p->q_row_stride + (int64_t)h * p->q_head_stride + d, bf);
            acc[d] = 0.f;
        }
        float m = -INFINITY, l = 0.f;
        int ntiles = (sk + TILE - 1) / TILE;
        if (p->is_causal) { /* prefill.cuh:750-752, clamped to the key length */
            const int lim = (r + 1 + TILE - 1) / TILE;
            if (lim < ntiles) ntiles = lim;
        }
        for (int j = 0; j < ntiles; ++j) {
            float s[TILE], pr[TILE];
            float tmax = -INFINITY;
            for (int i = 0; i < TILE; ++i) {
                const int t = j * TILE + i;
                if (t >= sk || (p->is_causal && t > r)) { /* prefill.cuh:393-421 */
                    s[i] = -INFINITY;
                    for (int d = 0; d < D; ++d) vbuf[i * D + d] = 0.f; /* zero-filled rows, :206-210 */
                    continue;
                }
                const int64_t ko = kv_row_offset(p, b, t, p->k_batch_stride, p->k_row_stride, p->k_cache_block_stride) +
                                   (int64_t)hk * p->k_head_stride;
                const int64_t vo = kv_row_offset(p, b, t, p->v_batch_stride, p->v_row_stride, p->v_cache_block_stride) +
                                   (int64_t)hk * p->v_head_stride;
                float dot = 0.f;
                for (int d = 0; d < D; ++d) {
                    dot += qrow[d] * ld(p->k_ptr, ko + d, bf);
                    vbuf[i * D + d] = ld(p->v_ptr, vo + d, bf);
                }
                s[i] = dot; /* unscaled, prefill.cuh:324-363 */
                if (dot > tmax) tmax = dot;
            }
            const float m_new = tmax > m ? tmax : m;             /* :454-462 */
            const float ms = (m_new == -INFINITY) ? 0.f : m_new; /* fully-masked-row guard */
            const float alpha = exp2f((m - ms) * c);             /* :461 */
            float psum = 0.f;
            for (int i = 0; i < TILE; ++i) {
                pr[i] = exp2f(s[i] * c - ms * c); /* :467-475 */
                psum += pr[i];                    /* un-rounded fp32 P, :476-482 */
            }
            l = l * alpha + psum;
            m = m_new;
            for (int d = 0; d < D; ++d) acc[d] *= alpha; /* :592-595 */
            for (int i = 0; i < TILE; ++i) {
                const float p16 = round_elt(pr[i], bf); /* :555-574 */
                if (p16 == 0.f) continue;
                for (int d = 0; d < D; ++d) acc[d] += p16 * vbuf[i * D + d];
            }
        }
        const float inv = (l == 0.f || l != l) ? 1.f : 1.f / l; /* :600-612 */
        for (int d = 0; d < D; ++d)
            st(p->o_ptr, oo + (int64_t)r * p->o_row_stride + (int64_t)h * p->o_head_stride + d, acc[d] * inv, bf);
    }
    free(vbuf);
    free(acc);
    free(qrow);
}

int mfa_oracle_prefill(const mfa_forward_params* p) {
    const int BH = p->batch * p->heads;
#pragma omp parallel for schedule(dynamic, 1)
    for (int bh = 0; bh < BH; ++bh) {
        const int b = bh / p->heads, h = bh % p->heads;
        int sq = p->seqlen_q, sk = p->seqlen_k;
        int64_t qo = (int64_t)b * p->q_batch_stride, oo = (int64_t)b * p->o_batch_stride;
        if (p->cu_seqlens_q) {
            sq = p->cu_seqlens_q[b + 1] - p->cu_seqlens_q[b];
            sk = p->cu_seqlens_k[b + 1] - p->cu_seqlens_k[b];
            qo = (int64_t)p->cu_seqlens_q[b] * p->q_row_stride;
            oo = (int64_t)p->cu_seqlens_q[b] * p->o_row_stride;
        }
        prefill_head(p, b, h, sq, sk, qo, oo);
    }
    return 0;
}

/* ---- decode (seqlen_q == 1), optional split-KV + combine -------------------------------------- */
/* One (b, h) over key tiles [t0, t1): the reference's 4 warps each own rows k = 4n + w of every tile. */
static void decode_range(const mfa_forward_params* p, int b, int h, int len, int t0, int t1, float* o_out,
                         float* lse_out) {
    const int D = p->head_dim, bf = p->is_bf16, hk = h / (p->heads / p->kv_heads);
    const float c = p->softmax_scale_log2;
    float mw[4], lw[4];
    float* ow = (float*)calloc((size_t)4 * D, sizeof(float));
    float* q = (float*)malloc(sizeof(float) * D);
    for (int w = 0; w < 4; ++w) { mw[w] = -INFINITY; lw[w] = 0.f; }
    for (int d = 0; d < D; ++d) q[d] = ld(p->q_ptr, (int64_t)b * p->q_batch_stride + (int64_t)h * p->q_head_stride + d, bf);
    for (int j = t0; j < t1; ++j) {
        for (int w = 0; w < 4; ++w) {
            float s[16];
            float tmax = -INFINITY;
            for (int n = 0; n < 16; ++n) {
                const int t = j * TILE + 4 * n + w; /* decode.cuh:285,297 */
                if (t >= len) { s[n] = -INFINITY; continue; } /* tail mask, :327-336 */
                const int64_t ko = kv_row_offset(p, b, t, p->k_batch_stride, p->k_row_stride, p->k_cache_block_stride) +
                                   (int64_t)hk * p->k_head_stride;
                float dot = 0.f;
                for (int d = 0; d < D; ++d) dot += q[d] * ld(p->k_ptr, ko + d, bf); /* :296-312 */
                s[n] = dot;
                if (dot > tmax) tmax = dot;
            }
            const float m_new = tmax > mw[w] ? tmax : mw[w];
            const float ms = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = exp2f((mw[w] - ms) * c); /* :367-383 */
            for (int d = 0; d < D; ++d) ow[w * D + d] *= alpha;
            float psum = 0.f;
            for (int n = 0; n < 16; ++n) {
                const int t = j * TILE + 4 * n + w;
                if (t >= len) continue;
                const float pr = exp2f(s[n] * c - ms * c); /* kept fp32 */
                psum += pr;
                const int64_t vo = kv_row_offset(p, b, t, p->v_batch_stride, p->v_row_stride, p->v_cache_block_stride) +
                                   (int64_t)hk * p->v_head_stride;
                for (int d = 0; d < D; ++d) ow[w * D + d] += pr * ld(p->v_ptr, vo + d, bf); /* :419-443 */
            }
            lw[w] = lw[w] * alpha + psum;
            mw[w] = m_new;
        }
    }
    /* block merge, decode.cuh:587-641 */
    float M = -INFINITY, L = 0.f;
    for (int w = 0; w < 4; ++w) if (mw[w] > M) M = mw[w];
    float f[4];
    for (int w = 0; w < 4; ++w) {
        f[w] = (mw[w] == -INFINITY) ? 0.f : exp2f((mw[w] - M) * c);
        L += lw[w] * f[w];
    }
    for (int d = 0; d < D; ++d) {
        float o = 0.f;
        for (int w = 0; w < 4; ++w) o += ow[w * D + d] * f[w];
        o_out[d] = L > 0.f ? o / L : 0.f;
    }
    *lse_out = L > 0.f ? M * p->softmax_scale + logf(L) : -INFINITY; /* :649 */
    free(ow);
    free(q);
}

static void decode_head(const mfa_forward_params* p, int b, int h) {
    const int D = p->head_dim, bf = p->is_bf16;
    const int S = p->num_splits < 1 ? 1 : p->num_splits;
    float* o = (float*)malloc(sizeof(float) * (size_t)D * S);
    float* lse = (float*)malloc(sizeof(float) * S);
    int len = p->seqlens_k ? p->seqlens_k[b] : p->seqlen_k;
    if (len < 0) len = 0;
    if (len > p->seqlen_k) len = p->seqlen_k;
    const int ntiles = (len + TILE - 1) / TILE;
    const int per = (ntiles + S - 1) / S; /* decode.cuh:26-30 */
    for (int s = 0; s < S; ++s) {
        int t0 = s * per, t1 = (s + 1) * per;
        if (t0 > ntiles) t0 = ntiles;
        if (t1 > ntiles) t1 = ntiles;
        decode_range(p, b, h, len, t0, t1, o + (size_t)s * D, lse + s);
        if (S > 1 && p->oaccum_ptr && p->softmax_lseaccum_ptr) {
            const int64_t slot = ((int64_t)s * p->batch + b) * p->heads + h;
            memcpy(p->oaccum_ptr + slot * D, o + (size_t)s * D, sizeof(float) * D);
            p->softmax_lseaccum_ptr[slot] = lse[s];
        }
    }
    /* combine, decode.cuh:718-747 (max-subtracted) */
    float M = -INFINITY, W = 0.f;
    for (int s = 0; s < S; ++s) if (lse[s] > M) M = lse[s];
    for (int s = 0; s < S; ++s) W += (lse[s] == -INFINITY) ? 0.f : expf(lse[s] - M);
    for (int d = 0; d < D; ++d) {
        float acc = 0.f;
        if (M != -INFINITY)
            for (int s = 0; s < S; ++s)
                if (lse[s] != -INFINITY) acc += o[(size_t)s * D + d] * expf(lse[s] - M);
        st(p->o_ptr, (int64_t)b * p->o_batch_stride + (int64_t)h * p->o_head_stride + d, M != -INFINITY ? acc / W : 0.f, bf);
    }
    if (p->softmax_lse_ptr) p->softmax_lse_ptr[(int64_t)b * p->heads + h] = M != -INFINITY ? M + logf(W) : -INFINITY;
    free(o);
    free(lse);
}

int mfa_oracle_decode(const mfa_forward_params* p) {
    const int BH = p->batch * p->heads;
#pragma omp parallel for schedule(dynamic, 1)
    for (int bh = 0; bh < BH; ++bh) decode_head(p, bh / p->heads, bh % p->heads);
    return 0;
}
