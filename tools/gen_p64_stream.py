#!/usr/bin/env python3
"""Generates mini-flash-attention_amd/csrc/mfa_prefill64_stream.inc: the instruction streams of prefill64_kernel
(mfa_prefill64.hip) as inline-asm blocks, one text per element type, with every vector register named physically.

    python tools/gen_p64_stream.py            # rewrites the .inc (committed; build.py does not run this)

The kernel's whole tile loop runs out of a RESERVED part of the register file that hipcc never sees as variables:
P64_INIT* takes the initial values as operands pinned to their home registers, every later block names the homes
literally and lists the whole reserved range as clobbered (so hipcc keeps nothing of its own there between
blocks), and P64_FINAL hands the results back as operands.  Between the blocks the kernel runs only scalar control
flow.  hipcc therefore neither allocates nor moves anything the streams touch: the order below IS the order on the
machine, and no compiler-generated copy can land between an MFMA and the use of its result.

Register map (one wave per SIMD, 512 registers: v0..v255 + a0..a255).  Reserved: v[VB : VB+NV) and a[AB : AB+NA); the
numbers below are RELATIVE to VB / AB.  hipcc allocates from register 0 upwards and keeps the low ones for its own values
(lane ids, the booleans it parks in VGPRs between blocks, ...):
    v[0:127]    S blocks: tile parity p, chain c: v[(2p+c)*32 + 16*kb + i]   (raw scores, then P packed in place)
    v[128:143]  V^T fragment ring (4 x 4)          v[144:151] K read addresses (per k-step)
    v[152:155]  V read addresses (per d & 3)        v[156:159] DMA lane constants k_go, k_gmax, v_go, v_gmax
    v[160:161]  l0, l1      v[162:163]  lt0, lt1    v[164:165] DMA address temporaries
    v[166:167]  row + hi of chain 0 / 1 (mask bound; 0x3fffffff without a right bound)      v168  4*h
    v[170:171]  -m*c of chain 0 / 1 (addend of the softmax fma)      v[172:173]  m0, m1 (max of the raw scores)
    v[174:179]  temporaries
    a[0:127]    O blocks: chain c, column block d: a[(4c+d)*16]
    a[128:191]  Q fragments: chain c, k-step ks: a[128 + (8c+ks)*4]
    a[192:207]  K fragment ring (4 x 4)
    s84..s99    scalar temporaries (clobbered)

Softmax of one element, in place:  x = fma(x, c, -m*c);  x = exp2(x)   with c = softmax_scale*log2(e) in fp32, as the
reference scales (prefill.cuh:452-483).  Q is NOT pre-multiplied by c: rounding c*q to 16 bits costs 2^-12 (fp16) / 2^-9
(bf16) relative per element, which the exponential amplifies with the score magnitude (measured: LSE off by 4e-3 in bf16,
O off by 2e-2 on fp16 inputs scaled by 6).  m moves only in the textbook blocks (first tile, or a tile whose row sums of P
exceed LIMIT): between them P may exceed 1 by up to LIMIT.

Blocks (P = parity of a tile's S buffer, C = chain); scalar operands are named in each block's _OPS macro:
    P64_INIT            operands -> home registers; O, l := 0           (a wave without rows: it only stages tiles)
    P64_INIT_X0         the same + phase X of tile 0 into buffer 0
    P64_FIRST0          the textbook softmax of tile 0 (sets m)
    P64_X1_FIRST0       phase X of tile 1 into buffer 1 with the textbook softmax of tile 0 in its gaps
    P64_STEADY          the steady-state loop: pairs of iterations (odd tile j, even tile j+1), each
                        [barrier] phase Y (P.V of tile j-1) | phase X (QK^T of tile j+1), the softmax of tile j and the
                        LDS reads / DMA pieces in the gaps between the MFMAs, fragment reads handed over between phases;
                        leaves when j >= jend (status 0) or when a tile's row sums fail the test (status 1)
    P64_X{P}_SM         phase X alone: scores of the tile k_rd points at into buffer P, softmax steps 32..63 of buffer P^1
    P64_Y{P}[_SM]       phase Y alone: O += V.P, P in buffer P, V tile v_rd points at [softmax steps 0..31 of buffer P^1]
    P64_SM2_{P}         softmax steps 32..63 of buffer P with nothing to hide under
    P64_MASK{P}         key > row + hi or key >= sk -> -inf on both chains of buffer P
    P64_CHECK           test of the two tile sums; passing chains: l += lt; status bit c = chain c failed
    P64_REDO{P}{C}      chain C of the tile in buffer P the textbook way: scores again from the K tile in the ring, mask,
                        new max, rescale of O and l, P
    P64_DMA_K / _V      the four 1-KiB pieces of one K / V tile as a burst
    P64_FINAL           home registers -> operands (O, l, m)
The phases outside the loop read their own first fragments (no hand-over), so any sequence of them is valid.
"""
import os

PF = 3          # fragments read ahead of their MFMAs (rings hold PF + 1)
TILE = 16384
RING = 3
V_RING = RING * TILE
NW = 4
NI = 16 // NW
LIMIT = 0x44800000  # 1024.0f
NEG_INF = 0xFF800000
NV, NA = 180, 208    # reserved VGPRs / AGPRs
VB, AB = 72, 48      # first reserved VGPR / AGPR


def S_BASE(p, c):
    return (2 * p + c) * 32


def VFR(k):
    return 128 + 4 * k


def KRD(ks):
    return 144 + ks


def VRD(d):
    return 152 + d


V_KGO, V_KGMAX, V_VGO, V_VGMAX = 156, 157, 158, 159


def L(c):
    return 160 + c


def LT(c):
    return 162 + c


VT = (164, 165)


def QHI(c):
    return 166 + c


H4 = 168


def MC(c):
    return 170 + c


def M(c):
    return 172 + c


def TMP(c, i):  # three temporaries per chain
    return 174 + 3 * c + i


def O_BASE(c, d):
    return (4 * c + d) * 16


def Q_BASE(c, ks):
    return 128 + (8 * c + ks) * 4


def KFR(k):
    return 192 + 4 * k


# scalar temporaries
(S_KOFF, S_VOFF, S_DSTK, S_DSTV, S_KDELTA, S_VDELTA, S_K16, S_V16, S_K64, S_V64, S_KNEXT, S_VNEXT, S_T0, S_T1, S_T2,
 S_T3) = range(84, 100)


def vr(lo, n=1):
    lo += VB
    return f"v{lo}" if n == 1 else f"v[{lo}:{lo + n - 1}]"


def ar(lo, n=1):
    lo += AB
    return f"a{lo}" if n == 1 else f"a[{lo}:{lo + n - 1}]"


class Stream:
    def __init__(self, f16):
        self.out = []
        self.mf = "v_mfma_f32_32x32x16_f16" if f16 else "v_mfma_f32_32x32x16_bf16"
        self.cvt = "v_cvt_pk_f16_f32" if f16 else "v_cvt_pk_bf16_f32"
        self.dma_t = 0
        self.lds_log = []
        self.ablate = set()  # developer timing builds: "dma", "sm", "lds" leave that part of the steady loop out

    def e(self, s):
        self.out.append(s)

    def pads(self):
        # results of MFMAs issued before this point readable by the VALU (18 wait states), registers written by the VALU
        # readable by the MFMAs
        self.e("s_nop 15")
        self.e("s_nop 7")

    # ---- softmax of one tile as 64 element steps u (u & 1: chain, u >> 1: element 16*kb + i): fma + exp in place now,
    # the row-sum add one element later, the pack of a finished pair (in place, word i/2) right behind its second add
    def sm_step(self, P, u):
        if "sm" in self.ablate:
            return
        ch, e = u & 1, u >> 1
        sb = S_BASE(P, ch)
        self.e(f"v_fma_f32 {vr(sb + e)}, {vr(sb + e)}, %[c], {vr(MC(ch))}")
        self.e(f"v_exp_f32 {vr(sb + e)}, {vr(sb + e)}")
        if u >= 2:
            e2 = e - 1
            self.e(f"v_add_f32 {vr(LT(ch))}, {vr(LT(ch))}, {vr(sb + e2)}")
            if e2 & 1:
                kb2, i2 = e2 >> 4, e2 & 15
                self.e(f"{self.cvt} {vr(sb + 16 * kb2 + (i2 >> 1))}, {vr(sb + e2 - 1)}, {vr(sb + e2)}")

    def sm_tail(self, P):
        if "sm" in self.ablate:
            return
        for ch in range(2):
            sb = S_BASE(P, ch)
            self.e(f"v_add_f32 {vr(LT(ch))}, {vr(LT(ch))}, {vr(sb + 31)}")
            self.e(f"{self.cvt} {vr(sb + 16 + 7)}, {vr(sb + 30)}, {vr(sb + 31)}")

    # LDS reads are logged in issue order (tag = (kind, phase sequence number, fragment)), so that a wait for a fragment
    # can be written as "all but the reads issued after it": lgkmcnt(N), N = reads younger than the fragment's last one
    def k_read(self, f, seq=0):
        kb, ks = f >> 3, f & 7
        self.lds_log.append(("K", seq, f))
        if "lds" not in self.ablate:
            self.e(f"ds_read_b128 {ar(KFR(f & PF), 4)}, {vr(KRD(ks))} offset:{kb * 32 * 256}")

    def v_read_half(self, f, half, seq=0):
        s16, d = f >> 2, f & 3
        self.lds_log.append(("V", seq, f))
        if "lds" not in self.ablate:
            self.e(f"ds_read_b64_tr_b16 {vr(VFR(f & PF) + 2 * half, 2)}, {vr(VRD(d))} offset:{s16 * 16 * 256 + half * 8 * 256}")

    def v_read(self, f, seq=0):
        self.v_read_half(f, 0, seq)
        self.v_read_half(f, 1, seq)

    def wait_frag(self, kind, seq, f, pad=False):
        """Everything up to fragment (kind, seq, f) has landed.  An MFMA must not follow the wait directly: a wait that
        really waited is passed a few cycles before the first dword is readable by the matrix core (measured: the first
        consumer lost that dword).  The streams put a slot's fillers between the two; bare phases pad with s_nop."""
        last = max(i for i, t in enumerate(self.lds_log) if t == (kind, seq, f))
        if "lds" in self.ablate:
            return
        self.e(f"s_waitcnt lgkmcnt({len(self.lds_log) - 1 - last})")
        if pad:
            self.e("s_nop 3")

    def dma_piece(self, pc):
        if "dma" in self.ablate:
            return
        vt = VT[self.dma_t & 1]
        self.dma_t += 1
        if pc < NI:
            go, gmax, off, dst, step, base, p = V_KGO, V_KGMAX, S_KOFF, S_DSTK, S_K16, "%[kbase]", pc
        else:
            go, gmax, off, dst, step, base, p = V_VGO, V_VGMAX, S_VOFF, S_DSTV, S_V16, "%[vbase]", pc - NI
        self.e(f"v_add_u32 {vr(vt)}, s{off}, {vr(go)}")
        self.e(f"v_min_u32 {vr(vt)}, {vr(vt)}, {vr(gmax)}")
        self.e(f"s_add_u32 m0, s{dst}, {p * NW * 1024}")
        self.e(f"s_add_u32 s{off}, s{off}, s{step}")  # (also the wait state between the M0 write and its use)
        self.e(f"global_load_lds_dwordx4 {vr(vt)}, {base}")

    def next_slot(self, dst, src):
        self.e(f"s_add_u32 s{dst}, {src}, {TILE}")
        self.e(f"s_cmp_eq_u32 s{dst}, {RING * TILE}")
        self.e(f"s_cselect_b32 s{dst}, 0, s{dst}")

    def mfma_x(self, PN, t):
        ch, f = t & 1, t >> 1
        kb, ks = f >> 3, f & 7
        sn = S_BASE(PN, ch) + 16 * kb
        c_op = "0" if ks == 0 else vr(sn, 16)
        self.e(f"{self.mf} {vr(sn, 16)}, {ar(KFR(f & PF), 4)}, {ar(Q_BASE(ch, ks), 4)}, {c_op}")

    def mfma_y(self, PP, t):
        ch, f = t & 1, t >> 1
        s16, d = f >> 2, f & 3
        kb, sh = s16 >> 1, s16 & 1
        o = O_BASE(ch, d)
        self.e(f"{self.mf} {ar(o, 16)}, {vr(VFR(f & PF), 4)}, {vr(S_BASE(PP, ch) + 16 * kb + 4 * sh, 4)}, {ar(o, 16)}")

    # ---- one iteration of the steady-state loop
    def iteration(self, P, fail_label, seq):
        """tile j = %[j] (parity P): P(j-1) and the destination of S(j+1) are in the buffer of parity P^1; seq: sequence
        number of its phase Y (the wait for that phase's first V fragment has been done by whoever came before)"""
        self.e(f"; ---- iteration, tile parity {P}")
        self.e("s_waitcnt vmcnt(0)")
        self.e("s_barrier")
        # ring slots: K(j+2) goes behind the slot K(j+1) is read from, V(j+1) two behind the slot V(j-1) is read from
        self.next_slot(S_KNEXT, "%[kslot]")
        self.e(f"s_add_u32 s{S_DSTK}, %[dst0], s{S_KNEXT}")
        self.e(f"s_sub_u32 s{S_KDELTA}, s{S_KNEXT}, %[kslot]")
        self.e(f"s_sub_u32 s{S_T0}, %[vslot], {TILE}")
        self.e("s_cmp_eq_u32 %[vslot], 0")
        self.e(f"s_cselect_b32 s{S_T0}, {(RING - 1) * TILE}, s{S_T0}")
        self.e(f"s_add_u32 s{S_DSTV}, %[dst0], s{S_T0}")
        self.e(f"s_add_u32 s{S_DSTV}, s{S_DSTV}, {V_RING}")
        self.next_slot(S_VNEXT, "%[vslot]")
        self.e(f"s_sub_u32 s{S_VDELTA}, s{S_VNEXT}, %[vslot]")
        self.e(f"s_add_u32 s{S_T0}, %[j], 2")
        self.e(f"s_mul_i32 s{S_KOFF}, s{S_T0}, s{S_K64}")
        self.e(f"s_add_u32 s{S_T0}, %[j], 1")
        self.e(f"s_mul_i32 s{S_VOFF}, s{S_T0}, s{S_V64}")
        # ---- phase Y: O^T += V^T.P^T of tile j-1 (slot t: chain t & 1, V fragment t >> 1 = 4*s16 + d).  The wait for a
        # fragment sits behind the MFMA of the slot before its first use (that slot's fillers separate it from the
        # consumer).  Fragment f + PF is read into the ring entry fragment f - 1 has left: its low half behind the first
        # MFMA of fragment f, its high half behind the second.
        for t in range(32):
            ch, f = t & 1, t >> 1
            self.mfma_y(P ^ 1, t)
            if ch == 1:
                if f < 15:
                    self.wait_frag("V", seq, f + 1)
                else:
                    self.wait_frag("K", seq + 1, 0)
            self.sm_step(P, t)
            if f + PF <= 15:
                self.v_read_half(f + PF, ch, seq)
            elif ch == 0:
                self.k_read(f + PF - 16, seq + 1)
            if ch == 1 and f >= 12:  # every V read through v_rd[f-12] is out: on to the next V tile
                self.e(f"v_add_u32 {vr(VRD(f - 12))}, s{S_VDELTA}, {vr(VRD(f - 12))}")
            if ch == 1 and f % 2 == 0:  # eight DMA pieces, behind every fourth MFMA
                self.dma_piece(f // 2)
        # ---- phase X: S^T = K.Q^T of tile j+1 into the buffer P^1 (slot t: chain t & 1, K fragment t >> 1 = 8*kb + ks)
        for t in range(32):
            ch, f = t & 1, t >> 1
            self.mfma_x(P ^ 1, t)
            if ch == 1:
                if f < 15:
                    self.wait_frag("K", seq + 1, f + 1)
                else:
                    self.wait_frag("V", seq + 2, 0)
            self.sm_step(P, 32 + t)
            if f + PF <= 15:
                if ch == 0:
                    self.k_read(f + PF, seq + 1)
            else:
                self.v_read_half(f + PF - 16, ch, seq + 2)
            if ch == 1 and f >= 8:  # every K read through k_rd[f-8] is out: on to the next K tile
                self.e(f"v_add_u32 {vr(KRD(f - 8))}, s{S_KDELTA}, {vr(KRD(f - 8))}")
        self.sm_tail(P)
        # ring state after the phases; then the test of the two tile sums: !(lt <= limit), NaN included
        self.e(f"s_mov_b32 %[kslot], s{S_KNEXT}")
        self.e(f"s_mov_b32 %[vslot], s{S_VNEXT}")
        self.e(f"v_cmp_nge_f32 vcc, 0x{LIMIT:x}, {vr(LT(0))}")
        self.e("s_nop 1")
        self.e(f"s_mov_b64 s[{S_T0}:{S_T1}], vcc")
        self.e(f"v_cmp_nge_f32 vcc, 0x{LIMIT:x}, {vr(LT(1))}")
        self.e("s_nop 1")
        self.e(f"s_or_b64 s[{S_T0}:{S_T1}], s[{S_T0}:{S_T1}], vcc")
        self.e(f"s_cmp_lg_u64 s[{S_T0}:{S_T1}], 0")
        self.e(f"s_cbranch_scc1 {fail_label}")
        for ch in range(2):
            self.e(f"v_add_f32 {vr(L(ch))}, {vr(L(ch))}, {vr(LT(ch))}")
        for ch in range(2):
            self.e(f"v_mov_b32 {vr(LT(ch))}, 0")
        self.e("s_add_u32 %[j], %[j], 1")

    def steady(self):
        self.e("; steady-state tile loop of prefill64_kernel (generated by tools/gen_p64_stream.py)")
        self.e("s_mov_b32 %[status], 0")
        self.e("s_cmp_ge_i32 %[j], %[jend]")
        self.e("s_cbranch_scc1 9f")
        self.e(f"s_lshl_b32 s{S_K64}, %[ksb], 6")
        self.e(f"s_lshl_b32 s{S_K16}, %[ksb], 4")
        self.e(f"s_lshl_b32 s{S_V64}, %[vsb], 6")
        self.e(f"s_lshl_b32 s{S_V16}, %[vsb], 4")
        self.pads()
        for f in range(PF):  # the first V fragments of phase Y of the first iteration
            self.v_read(f, 0)
        self.wait_frag("V", 0, 0, pad=True)
        self.e("1:")
        self.iteration(1, "7f", 0)
        self.iteration(0, "7f", 2)
        # (the loop closes here: the reads still in flight are the first V fragments of sequence 4 == 0 of the next trip)
        assert self.lds_log[-6:] == [("V", 4, 0)] * 2 + [("V", 4, 1)] * 2 + [("V", 4, 2)] * 2
        self.e("s_cmp_lt_i32 %[j], %[jend]")
        self.e("s_cbranch_scc1 1b")
        self.e("s_branch 9f")
        self.e("7:")
        self.e("s_mov_b32 %[status], 1")
        self.e("9:")
        self.e("s_waitcnt lgkmcnt(0)")  # fragments read ahead for an iteration that will not run here
        self.pads()
        return self.out

    # ---- the self-contained phases used outside the steady-state loop; `fill`: extra instructions to spread over the
    # gaps (a list, consumed in order)
    def phase_x(self, PN, sm, fill=None):
        fill = list(fill or [])
        share = [fill[len(fill) * t // 32:len(fill) * (t + 1) // 32] for t in range(32)]
        self.pads()
        self.next_slot(S_KNEXT, "%[kslot]")
        self.e(f"s_sub_u32 s{S_KDELTA}, s{S_KNEXT}, %[kslot]")
        for f in range(PF):
            self.k_read(f)
        self.wait_frag("K", 0, 0, pad=True)
        for t in range(32):
            ch, f = t & 1, t >> 1
            self.mfma_x(PN, t)
            if ch == 1 and f < 15:
                self.wait_frag("K", 0, f + 1, pad=not sm and len(share[t]) < 3)
            if sm:
                self.sm_step(PN ^ 1, 32 + t)
            self.out += share[t]
            if ch == 0 and f + PF <= 15:
                self.k_read(f + PF)
            if ch == 1 and f >= 8:
                self.e(f"v_add_u32 {vr(KRD(f - 8))}, s{S_KDELTA}, {vr(KRD(f - 8))}")
        if sm:
            self.sm_tail(PN ^ 1)
        self.e(f"s_mov_b32 %[kslot], s{S_KNEXT}")
        self.pads()
        return self.out

    def phase_y(self, PP, sm):
        self.pads()
        self.next_slot(S_VNEXT, "%[vslot]")
        self.e(f"s_sub_u32 s{S_VDELTA}, s{S_VNEXT}, %[vslot]")
        for f in range(PF):
            self.v_read(f)
        self.wait_frag("V", 0, 0, pad=True)
        for t in range(32):
            ch, f = t & 1, t >> 1
            self.mfma_y(PP, t)
            if ch == 1 and f < 15:
                self.wait_frag("V", 0, f + 1, pad=not sm)
            if sm:
                self.sm_step(PP ^ 1, t)
            if f + PF <= 15:
                self.v_read_half(f + PF, ch)
            if ch == 1 and f >= 12:
                self.e(f"v_add_u32 {vr(VRD(f - 12))}, s{S_VDELTA}, {vr(VRD(f - 12))}")
        self.e(f"s_mov_b32 %[vslot], s{S_VNEXT}")
        self.pads()
        return self.out

    def sm_second_half(self, P):
        self.pads()
        for u in range(32, 64):
            self.sm_step(P, u)
        self.sm_tail(P)
        self.e("s_nop 4")
        return self.out

    # ---- pieces run once per workgroup or rarely -------------------------------------------------------------------------
    def mask_chain(self, P, ch):
        """key > row + hi or key >= sk -> -inf.  %[skm1] = sk - 1, %[j64] = 64 * tile index"""
        sb = S_BASE(P, ch)
        t0, t1 = TMP(ch, 0), TMP(ch, 1)
        self.e(f"v_min_i32 {vr(t0)}, %[skm1], {vr(QHI(ch))}")
        self.e(f"v_subrev_u32 {vr(t0)}, %[j64], {vr(t0)}")
        self.e(f"v_sub_u32 {vr(t0)}, {vr(t0)}, {vr(H4)}")  # keys at tile offsets <= t0 stay
        self.e(f"v_mov_b32 {vr(t1)}, 0x{NEG_INF:x}")
        for kb in range(2):
            for i in range(16):
                k = 32 * kb + (i & 3) + 8 * (i >> 2)
                self.e(f"v_cmp_gt_i32 vcc, {k}, {vr(t0)}")
                self.e(f"v_cndmask_b32 {vr(sb + 16 * kb + i)}, {vr(sb + 16 * kb + i)}, {vr(t1)}, vcc")

    def exact_softmax(self, P, ch, first):
        """the textbook update of chain ch for the tile whose raw scores are in buffer P (reference prefill.cuh:452-483): new
        max, rescale of O and l, P = exp2((score - m_new)*c) packed in place, l += sum.  Returns the instruction list."""
        out, self.out = self.out, []
        sb = S_BASE(P, ch)
        t0, t1, t2 = TMP(ch, 0), TMP(ch, 1), TMP(ch, 2)
        self.e(f"v_max3_f32 {vr(t0)}, {vr(sb)}, {vr(sb + 1)}, {vr(sb + 2)}")
        for i in range(3, 31, 2):
            self.e(f"v_max3_f32 {vr(t0)}, {vr(t0)}, {vr(sb + i)}, {vr(sb + i + 1)}")
        self.e(f"v_max_f32 {vr(t0)}, {vr(t0)}, {vr(sb + 31)}")
        self.e(f"v_mov_b32 {vr(t1)}, {vr(t0)}")
        self.e("s_nop 1")
        self.e(f"v_permlane32_swap_b32 {vr(t0)}, {vr(t1)}")  # t0 = {lo, lo}, t1 = {hi, hi}
        self.e("s_nop 1")
        self.e(f"v_max_f32 {vr(t0)}, {vr(t0)}, {vr(t1)}")  # max of the raw scores over the row's 64 keys
        if first:
            self.e(f"v_mov_b32 {vr(M(ch))}, {vr(t0)}")
        else:
            self.e(f"v_max_f32 {vr(t0)}, {vr(t0)}, {vr(M(ch))}")  # m_new (a fully masked tile leaves m)
            self.e(f"v_sub_f32 {vr(t1)}, {vr(M(ch))}, {vr(t0)}")
            self.e(f"v_mul_f32 {vr(t1)}, %[c], {vr(t1)}")
            self.e(f"v_exp_f32 {vr(t1)}, {vr(t1)}")  # alpha = 2^((m_old - m_new)*c)
            self.e(f"v_mov_b32 {vr(M(ch))}, {vr(t0)}")
            self.e(f"v_mul_f32 {vr(L(ch))}, {vr(L(ch))}, {vr(t1)}")
            for d in range(4):
                for i in range(16):
                    a = O_BASE(ch, d) + i
                    self.e(f"v_accvgpr_read_b32 {vr(t2)}, {ar(a)}")
                    self.e(f"v_mul_f32 {vr(t2)}, {vr(t2)}, {vr(t1)}")
                    self.e(f"v_accvgpr_write_b32 {ar(a)}, {vr(t2)}")
        self.e(f"v_mul_f32 {vr(MC(ch))}, %[c], {vr(t0)}")
        self.e(f"v_sub_f32 {vr(MC(ch))}, 0, {vr(MC(ch))}")  # -m*c
        for e_ in range(32):
            self.e(f"v_fma_f32 {vr(sb + e_)}, {vr(sb + e_)}, %[c], {vr(MC(ch))}")
        for e_ in range(32):
            self.e(f"v_exp_f32 {vr(sb + e_)}, {vr(sb + e_)}")
        self.e(f"v_mov_b32 {vr(t2)}, 0")
        for e_ in range(32):
            self.e(f"v_add_f32 {vr(t2)}, {vr(t2)}, {vr(sb + e_)}")
        for kb in range(2):
            for i in range(8):
                b = sb + 16 * kb
                self.e(f"{self.cvt} {vr(b + i)}, {vr(b + 2 * i)}, {vr(b + 2 * i + 1)}")
        self.e(f"v_add_f32 {vr(L(ch))}, {vr(L(ch))}, {vr(t2)}")
        self.e(f"v_mov_b32 {vr(LT(ch))}, 0")
        lines, self.out = self.out, out
        return lines

    def first_lines(self, P):
        """the first tile's textbook softmax, the two chains' instructions alternating (independent dependency chains)"""
        a, b = self.exact_softmax(P, 0, True), self.exact_softmax(P, 1, True)
        return [x for pair in zip(a, b) for x in pair]

    def mask_block(self, P):
        self.pads()
        self.mask_chain(P, 0)
        self.mask_chain(P, 1)
        self.e("s_nop 4")
        return self.out

    def first_block(self, P):
        self.pads()
        self.out += self.first_lines(P)
        self.e("s_nop 4")
        return self.out

    def x_first_block(self):
        """phase X of tile 1 (buffer 1) with the textbook softmax of tile 0 (buffer 0) in its gaps"""
        return self.phase_x(1, False, fill=self.first_lines(0))

    def check_block(self):
        """status bit c = chain c failed the test of its tile sum; passing chains: l += lt, lt = 0"""
        self.e("s_mov_b32 %[status], 0")
        for ch in range(2):
            self.e(f"v_cmp_nge_f32 vcc, 0x{LIMIT:x}, {vr(LT(ch))}")
            self.e("s_nop 1")
            self.e("s_cmp_lg_u64 vcc, 0")
            self.e(f"s_cbranch_scc1 {ch + 1}f")
            self.e(f"v_add_f32 {vr(L(ch))}, {vr(L(ch))}, {vr(LT(ch))}")
            self.e(f"v_mov_b32 {vr(LT(ch))}, 0")
            self.e(f"s_branch {ch + 3}f")
            self.e(f"{ch + 1}:")
            self.e(f"s_or_b32 %[status], %[status], {1 << ch}")
            self.e(f"{ch + 3}:")
        return self.out

    def redo_block(self, P, ch):
        """%[kdelta]: byte offset of the tile's K ring slot relative to the slot k_rd points at"""
        self.pads()
        t2 = TMP(ch, 2)
        for kb in range(2):
            for ks in range(8):
                sn = S_BASE(P, ch) + 16 * kb
                self.e(f"v_add_u32 {vr(t2)}, %[kdelta], {vr(KRD(ks))}")
                self.e(f"ds_read_b128 {ar(KFR(0), 4)}, {vr(t2)} offset:{kb * 32 * 256}")
                self.e("s_waitcnt lgkmcnt(0)")
                self.e("s_nop 3")
                c_op = "0" if ks == 0 else vr(sn, 16)
                self.e(f"{self.mf} {vr(sn, 16)}, {ar(KFR(0), 4)}, {ar(Q_BASE(ch, ks), 4)}, {c_op}")
        self.pads()
        self.mask_chain(P, ch)
        self.out += self.exact_softmax(P, ch, False)
        self.e("s_nop 4")
        return self.out

    def dma_block(self, is_v):
        """%[off]: byte offset of the tile's first row, %[dst]: LDS byte address of this wave's first piece in the slot,
        %[step]: 16 rows in bytes, %[base]: the K / V base of the (batch, head)"""
        go, gmax = (V_VGO, V_VGMAX) if is_v else (V_KGO, V_KGMAX)
        self.e(f"s_mov_b32 s{S_T0}, %[off]")
        for p in range(NI):
            vt = VT[p & 1]
            self.e(f"v_add_u32 {vr(vt)}, s{S_T0}, {vr(go)}")
            self.e(f"v_min_u32 {vr(vt)}, {vr(vt)}, {vr(gmax)}")
            self.e(f"s_add_u32 m0, %[dst], {p * NW * 1024}")
            self.e(f"s_add_u32 s{S_T0}, s{S_T0}, %[step]")
            self.e(f"global_load_lds_dwordx4 {vr(vt)}, %[base]")
        return self.out

    def init_lines(self):
        lines = [f"v_accvgpr_write_b32 {ar(i)}, 0" for i in range(128)]
        lines += [f"v_mov_b32 {vr(r)}, 0" for r in (L(0), L(1), LT(0), LT(1), M(0), M(1), MC(0), MC(1))]
        return lines

    def init_block(self):
        self.out += self.init_lines()
        self.e("s_nop 4")
        return self.out

    def init_x0_block(self):
        """phase X of tile 0 into buffer 0, the zeroing of O and l in its gaps"""
        return self.phase_x(0, False, fill=self.init_lines())


# ---- operand / clobber lists (C++ names of mfa_prefill64.hip) -------------------------------------------------------------
S_TMP = [f"s{i}" for i in range(84, 100)]
C_OP = '[c] "s"(c_log2)'


def all_regs(exclude=()):
    ex = set(exclude)
    return ([f"v{i}" for i in range(VB, VB + NV) if f"v{i}" not in ex] +
            [f"a{i}" for i in range(AB, AB + NA) if f"a{i}" not in ex])


def clob(regs):
    return ", ".join(f'"{r}"' for r in regs + S_TMP + ["vcc", "scc", "memory"])


def rng(kind, lo, n):
    lo += VB if kind == "v" else AB
    return [f"{kind}{i}" for i in range(lo, lo + n)]


def pin(kind, lo, n, mode=""):
    lo += VB if kind == "v" else AB
    return f'"{mode}{{{kind}[{lo}:{lo + n - 1}]}}"'


def emit_block(fh, name, lines_of, outs, ins, exclude=()):
    n = 0
    for suffix, f16 in (("F16", True), ("BF16", False)):
        lines = lines_of(Stream(f16))
        fh.write(f"#define {name}_{suffix} \\\n")
        for ln in lines:
            fh.write(f'    "{ln}\\n\\t" \\\n')
        fh.write('    ""\n')
        n = sum(1 for ln in lines if not ln.startswith(';') and not ln.endswith(':'))
    fh.write(f"// {name}: {n} instructions\n")
    fh.write(f"#define {name}_OPS : " + ", ".join(outs) + " : " + ", ".join(ins) + " : " + clob(all_regs(exclude)) + "\n\n")


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "mini-flash-attention_amd", "csrc", "mfa_prefill64_stream.inc")
    KS, VS = '[kslot] "+s"(k_slot_off)', '[vslot] "+s"(v_slot_off)'
    steady_outs = ['[j] "+s"(j)', KS, VS, '[status] "=&s"(status)']
    steady_ins = ['[jend] "s"(jend)', '[kbase] "s"(kbase)', '[vbase] "s"(vbase)', '[ksb] "s"(k_sb)', '[vsb] "s"(v_sb)',
                  '[dst0] "s"(dma_dst0)', C_OP]
    with open(path, "w") as fh:
        fh.write("// GENERATED by tools/gen_p64_stream.py -- do not edit.  The instruction streams of prefill64_kernel as inline-asm\n")
        fh.write("// blocks (one text per element type) with their operand lists; register map in the generator's docstring.\n")
        fh.write("// Use:  asm volatile(NAME_F16 NAME_OPS);  inside prefill64_kernel (the operand names are its variables).\n\n")
        # operands -> home registers (pinned inputs; everything else of the reserved range is zeroed or clobbered)
        q_ins = [pin("a", Q_BASE(c, ks), 4) + f"(Q[{c}][{ks}])" for c in range(2) for ks in range(8)]
        fixed_ins = q_ins + [pin("v", VRD(0), 4) + "(v_rd)", pin("v", V_KGO, 4) + "(dma_c)", pin("v", QHI(0), 3) + "(row_c)"]
        fixed_regs = rng("a", 128, 64) + rng("v", VRD(0), 4) + rng("v", V_KGO, 4) + rng("v", QHI(0), 3)
        krd_regs = rng("v", KRD(0), 8)
        emit_block(fh, "P64_INIT", lambda st: st.init_block(), [], fixed_ins + [pin("v", KRD(0), 8) + "(k_rd)"],
                   exclude=fixed_regs + krd_regs)
        emit_block(fh, "P64_INIT_X0", lambda st: st.init_x0_block(), [KS, pin("v", KRD(0), 8, "+") + "(k_rd)"], fixed_ins,
                   exclude=fixed_regs + krd_regs)
        emit_block(fh, "P64_FIRST0", lambda st: st.first_block(0), [], [C_OP])
        emit_block(fh, "P64_X1_FIRST0", lambda st: st.x_first_block(), [KS], [C_OP])
        emit_block(fh, "P64_STEADY", lambda st: st.steady(), steady_outs, steady_ins)
        for tag in ("dma", "sm", "lds"):  # timing-only variants of the steady loop (results are wrong): MFA_P64_DEBUG bits 2..4
            def ablated(st, tag=tag):
                st.ablate = {tag}
                return st.steady()
            emit_block(fh, f"P64_STEADY_NO_{tag.upper()}", ablated, steady_outs, steady_ins)
        for pn in range(2):
            emit_block(fh, f"P64_X{pn}_SM", lambda st, pn=pn: st.phase_x(pn, True), [KS], [C_OP])
            emit_block(fh, f"P64_Y{pn}", lambda st, pn=pn: st.phase_y(pn, False), [VS], [])
            emit_block(fh, f"P64_Y{pn}_SM", lambda st, pn=pn: st.phase_y(pn, True), [VS], [C_OP])
            emit_block(fh, f"P64_SM2_{pn}", lambda st, pn=pn: st.sm_second_half(pn), [], [C_OP])
            emit_block(fh, f"P64_MASK{pn}", lambda st, pn=pn: st.mask_block(pn), [], ['[skm1] "s"(skm1)', '[j64] "s"(j64)'])
            for ch in range(2):
                emit_block(fh, f"P64_REDO{pn}{ch}", lambda st, pn=pn, ch=ch: st.redo_block(pn, ch), [],
                           ['[skm1] "s"(skm1)', '[j64] "s"(j64)', '[kdelta] "s"(kdelta)', C_OP])
        emit_block(fh, "P64_CHECK", lambda st: st.check_block(), ['[status] "=&s"(status)'], [])
        for nm, isv in (("P64_DMA_K", False), ("P64_DMA_V", True)):
            emit_block(fh, nm, lambda st, isv=isv: st.dma_block(isv), [],
                       ['[off] "s"(dma_off)', '[dst] "s"(dma_dst)', '[step] "s"(dma_step)', '[base] "s"(dma_base)'])
        # home registers -> operands: an empty statement whose outputs are pinned to the homes
        o_outs = [pin("a", O_BASE(c, d), 16, "=") + f"(O[{c}][{d}])" for c in range(2) for d in range(4)]
        fh.write('#define P64_FINAL_F16 ""\n#define P64_FINAL_BF16 ""\n')
        fh.write("#define P64_FINAL_OPS : " + ", ".join(o_outs + [pin("v", L(0), 2, "=") + "(l2)", pin("v", M(0), 2, "=") + "(m2)"]) +
                 " : : \"memory\"\n")
    print("wrote", os.path.relpath(path, root))


if __name__ == "__main__":
    main()
