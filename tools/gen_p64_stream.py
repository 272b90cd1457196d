#!/usr/bin/env python3
"""Generates mini-flash-attention_amd/csrc/mfa_prefill64_stream.inc: the instruction streams of prefill64_kernel
(mfa_prefill64.hip) as inline-asm blocks, one text per element type, with every vector register named physically.

    python tools/gen_p64_stream.py            # rewrites the .inc (committed; build.py does not run this, but
                                              # tests/test_p64_stream_cpu.py fails when the two disagree)
    python tools/gen_p64_stream.py --out F    # writes F instead
    python tools/gen_p64_stream.py --dev      # + mfa_prefill64_stream_dev.inc: timing-only loop variants for -DMFA_DEV_P64 builds

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
    v[128:143]  V^T fragment ring (4 x 4)          v[144:151] K read addresses of ring slot 0 (per k-step)
    v[152:155]  V read addresses of ring slot 0 (per d & 3)           v[156:157] DMA lane offsets k_go, v_go
    v[158:159], v[164:165]  DMA lane offsets of the Q staging, by piece & 3
    v[160:161]  l0, l1      v[162:163]  lt0, lt1 (row sums of the tile in flight)
    v[166:167]  row + hi of chain 0 / 1 (mask bound; 0x3fffffff without a right bound)      v168  4*h
    v169        paged K/V (P64_STEADY_PG): the sequence's block-table row, lane i = entry i (P64_SETTAB)
    v[170:171]  -m*c of chain 0 / 1 (addend of the softmax fma)      v[172:173]  m0, m1 (max of the raw scores)
    v[174:179]  temporaries
    a[0:127]    O blocks: chain c, column block d: a[(4c+d)*16]
    a[128:191]  Q fragments: chain c, k-step ks: a[128 + (8c+ks)*4]
    a[192:207]  K fragment ring (4 x 4)
    s84, s85    scalar temporaries (clobbered)

LDS: K ring (3 tiles of 16 KiB) then V ring (3 tiles); tile t of either lives in slot t % 3.  The read-address registers
always hold slot 0; the steady-state loop is unrolled over the six (slot, S-buffer parity) combinations and selects slots
through the instructions' immediate offsets, the other blocks take the slot's byte offset as a scalar operand.

K/V tiles arrive by LDS-DMA through buffer descriptors (buffer_load_dwordx4 ... lds): lane offset k_go / v_go (row
4*wave + lane/16 of a 16-row piece and the 16-byte chunk the swizzled image wants at this lane's position), scalar offset
= the piece's first row.  The descriptors end at the last row the workgroup needs, so a piece past the end (ragged last
tile, or a tile nobody will read) writes zeros and moves no data.  %[koff] / %[voff] are running: every block that stages a
tile advances them by the tile's 64 rows, and every wave stages K(j+2) and V(j+1) in iteration j, in that order.

Softmax of one element, in place:  x = fma(x, c, -m*c);  x = exp2(x)   with c = softmax_scale*log2(e) in fp32, as the
reference scales (prefill.cuh:452-483).  Q is NOT pre-multiplied by c: rounding c*q to 16 bits costs 2^-12 (fp16) / 2^-9
(bf16) relative per element, which the exponential amplifies with the score magnitude (measured: LSE off by 4e-3 in bf16,
O off by 2e-2 on fp16 inputs scaled by 6).  m moves only in the textbook blocks (first tile, or a tile whose row sums of P
exceed LIMIT): between them P may exceed 1 by up to LIMIT.

One wave issues one instruction per 4 cycles (v_exp_f32: 8; an MFMA holds the issue port for 8 of its 32): a gap between
two MFMAs hides at most five other instructions, one of them a v_exp_f32.  The steady-state stream is written against that
budget: per gap one softmax element (fma, exp, row-sum add, every other gap a pack) and on average 1.4 others.

Blocks (P = parity of a tile's S buffer, C = chain); scalar operands are named in each block's _OPS macro:
    P64_SETUP           lane constants (LDS read addresses, DMA lane offsets, 4h) -> home registers, once per workgroup
    P64_DMA_Q           the wave's Q rows of a work item: global memory -> its LDS Q buffer (asynchronous: vmcnt)
    P64_Q_LDS           Q fragments: LDS Q buffer -> home registers
    P64_X0              a new work item: phase X of tile 0 into buffer 0; O, l := 0 and the mask bounds in its gaps
    P64_FIRST0          the textbook softmax of tile 0 (sets m)
    P64_X1_FIRST0       phase X of tile 1 into buffer 1 with the textbook softmax of tile 0 in its gaps
    P64_STEADY          the steady-state loop over tiles j = %[j] .. %[jend]-1, entered at any j (%[entry] = j % 6): per
                        iteration [barrier] phase Y (P.V of tile j-1) | phase X (QK^T of tile j+1), the softmax of tile j,
                        the LDS reads and the DMA pieces of K(j+2), V(j+1) in the gaps between the MFMAs, fragment reads
                        handed over between phases; leaves at jend (status 0) or when a tile's row sums fail the test
                        (status 1, tile j's phases done, its sums not yet accepted)
    P64_STEADY_PG       the same loop over a PAGED K/V cache (pages of 64 << %[tps] keys, at most 64 per sequence): a tile's
                        descriptor (page base = pool + table entry * block bytes, records = the page's rows that exist) and
                        running offset are rebuilt on the scalar side, one instruction per gap of phase X, for the tile
                        the NEXT iteration stages; the descriptors live in s[88:91] (K) and s[92:95] (V)
    P64_SETTAB          a work item's block-table row -> its home register
    P64_X{P}_SM         phase X alone: scores of the K tile in slot %[kslot] into buffer P, softmax steps 32..63 of buffer P^1
    P64_Y{P}[_SM]       phase Y alone: O += V.P, P in buffer P, V tile in slot %[vslot] [softmax steps 0..31 of buffer P^1]
    P64_LAST{P}[_M]     a wave's last tile (scores in buffer P) in one piece: phase Y of the tile before with the first half
                        of its softmax, the second half, the test of the sums; _M: its mask applied on the way
    P64_SM2_{P}         softmax steps 32..63 of buffer P with nothing to hide under
    P64_MASK{P}         key > row + hi or key >= sk -> -inf on both chains of buffer P
    P64_CHECK           test of the two tile sums; passing chains: l += lt; status bit c = chain c failed
    P64_REDO{P}{C}      chain C of the tile in buffer P the textbook way: scores again from the K tile in slot %[kslot], mask,
                        new max, rescale of O and l, P
    P64_DMA_K / _V      the four 1-KiB pieces of one K / V tile as a burst
    P64_EPILOGUE        O / l -> global memory: 1/l, pack, rows through this wave's LDS staging area, buffer stores
    P64_FINAL           home registers -> operands (l, m: for the LSE)
The phases outside the loop read their own first fragments (no hand-over), so any sequence of them is valid.
"""
import os
import sys

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


V_KGO, V_VGO = 156, 157
QGO = (158, 159, 164, 165)  # Q staging lane offsets, by piece & 3


def L(c):
    return 160 + c


def LT(c):
    return 162 + c


def QHI(c):
    return 166 + c


H4 = 168
TAB = 169  # paged K/V: block-table row of the work item's sequence
K_SRD, V_SRD = 88, 92  # P64_STEADY_PG: physical SGPR quads of the running K / V descriptors


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


S_T0 = 84  # scalar temporary
# Non-temporal O stores ("o") and Q requests ("q"): both stream through once, and L2 is exactly as large as the K/V of the (batch,
# head) pairs an XCD has in flight.  Same-box A/B (profiles/r03_ab_p64_designs.txt): S=512 causal -4.7 %, S=1024 causal -0.6 %, the rest
# unchanged.  P64_NT overrides for a developer A/B.
NT = os.environ.get("P64_NT", "oq")


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
        self.lds_log = []
        self.ablate = set()  # developer timing builds: "dma", "sm", "lds" leave that part of the steady loop out
        self.paged = False   # P64_STEADY_PG

    def page_update(self, is_v, ahead):
        """Descriptor and running offset of the V / K tile j + ahead of a paged cache (s84..s87 temporaries): page index
        (clamped to the sequence's last page), table entry out of the TAB register, 64-bit page base, records = rows of
        the page that exist x row pitch (rows past the last key read as zeros, as the dense descriptors arrange), offset =
        the tile's first row inside the page."""
        srd = V_SRD if is_v else K_SRD
        pool = "vpool" if is_v else "kpool"
        off = "%[voff]" if is_v else "%[koff]"
        return [
            [f"s_add_u32 s84, %[j], {ahead}"],
            ["s_lshr_b32 s85, s84, %[tps]"],
            ["s_min_u32 s85, s85, %[maxpg]"],
            [f"v_readlane_b32 s86, {vr(TAB)}, s85"],
            ["s_mul_hi_u32 s87, s86, %[blk]"],
            ["s_mul_i32 s86, s86, %[blk]"],
            [f"s_add_u32 s{srd}, %[{pool}lo], s86", f"s_addc_u32 s{srd + 1}, %[{pool}hi], s87"],
            ["s_lshl_b32 s86, s85, %[pshift]"],
            ["s_sub_u32 s86, %[kvrows], s86"],
            ["s_min_u32 s86, s86, %[pagerows]"],
            [f"s_mul_i32 s{srd + 2}, s86, %[sb]"],
            ["s_and_b32 s86, s84, %[tppm1]"],
            ["s_lshl_b32 s86, s86, 6"],
            [f"s_mul_i32 {off}, s86, %[sb]"],
        ]

    def e(self, s):
        self.out.append(s)

    def pads(self):
        # results of MFMAs issued before this point readable by the VALU (18 wait states), registers written by the VALU
        # readable by the MFMAs
        self.e("s_nop 15")
        self.e("s_nop 7")

    # ---- softmax of one tile as 64 element steps u (u & 1: chain, u >> 1: element 16*kb + i): fma + exp in place now,
    # the row-sum add one element later (the first add of a tile writes lt = x0 + x1), the pack of a finished pair (in
    # place, word i/2) right behind its second add
    def sm_step(self, P, u, masked=False):
        if "sm" in self.ablate or ("smx" in self.ablate and u >= 32) or ("smy" in self.ablate and u < 32):
            return
        ch, e = u & 1, u >> 1
        if masked:  # (last_block) key > row + hi or key >= sk -> -inf, ahead of the element's fma
            k = 32 * (e >> 4) + ((e & 15) & 3) + 8 * ((e & 15) >> 2)
            self.e(f"v_cmp_gt_i32 vcc, {k}, {vr(TMP(ch, 0))}")
            self.e(f"v_cndmask_b32 {vr(S_BASE(P, ch) + e)}, {vr(S_BASE(P, ch) + e)}, {vr(TMP(0, 1))}, vcc")

        x = vr(S_BASE(P, ch) + e)
        self.e(f"v_fma_f32 {x}, {x}, %[c], {vr(MC(ch))}")  # (issuing the fma a step ahead of its exp: measured, no gain)
        self.e(f"v_exp_f32 {x}, {x}")
        if u >= 2:
            self.sm_sum(P, ch, e - 1)

    def sm_sum(self, P, ch, e2):
        sb = S_BASE(P, ch)
        if e2 == 1:
            self.e(f"v_add_f32 {vr(LT(ch))}, {vr(sb)}, {vr(sb + 1)}")
        elif e2 > 1:
            self.e(f"v_add_f32 {vr(LT(ch))}, {vr(LT(ch))}, {vr(sb + e2)}")
        if e2 & 1 and "cvt" not in self.ablate:
            kb2, i2 = e2 >> 4, e2 & 15
            self.e(f"{self.cvt} {vr(sb + 16 * kb2 + (i2 >> 1))}, {vr(sb + e2 - 1)}, {vr(sb + e2)}")

    def sm_tail(self, P):
        if "sm" in self.ablate:
            return
        for ch in range(2):
            self.sm_sum(P, ch, 31)

    # LDS reads are logged in issue order (tag = (kind, phase sequence number, fragment)), so that a wait for a fragment
    # can be written as "all but the reads issued after it": lgkmcnt(N), N = reads younger than the fragment's last one.
    # slot: ring slot as an immediate (the address registers hold slot 0)
    def k_read(self, f, seq=0, slot=0):
        kb, ks = f >> 3, f & 7
        self.lds_log.append(("K", seq, f))
        if "lds" not in self.ablate:
            self.e(f"ds_read_b128 {ar(KFR(f & PF), 4)}, {vr(KRD(ks))} offset:{slot * TILE + kb * 32 * 256}")

    def v_read_half(self, f, half, seq=0, slot=0):
        s16, d = f >> 2, f & 3
        self.lds_log.append(("V", seq, f))
        if "lds" not in self.ablate:
            self.e(f"ds_read_b64_tr_b16 {vr(VFR(f & PF) + 2 * half, 2)}, {vr(VRD(d))} "
                   f"offset:{slot * TILE + s16 * 16 * 256 + half * 8 * 256}")

    def v_read(self, f, seq=0, slot=0):
        self.v_read_half(f, 0, seq, slot)
        self.v_read_half(f, 1, seq, slot)

    def wait_frag(self, kind, seq, f, pad=False):
        """Everything up to fragment (kind, seq, f) has landed.  An MFMA must not follow the wait directly: a wait that
        really waited is passed a few cycles before the first dword is readable by the matrix core (measured: the first
        consumer lost that dword).  The streams put a slot's fillers between the two; bare phases pad with s_nop."""
        last = max(i for i, t in enumerate(self.lds_log) if t == (kind, seq, f))
        if "lds" in self.ablate or "wait" in self.ablate:
            return
        self.e(f"s_waitcnt lgkmcnt({len(self.lds_log) - 1 - last})")
        if pad:
            self.e("s_nop 3")

    def filler(self):
        for tag in self.ablate:
            if tag.startswith("fill"):
                for n in range(int(tag[4:])):
                    k = getattr(self, "_fk", 0)
                    self._fk = (k + 1) % 6
                    self.e(f"v_add_f32 {vr(TMP(0, 0) + k)}, {vr(TMP(0, 0) + k)}, {vr(MC(0))}")
            if tag.startswith("fexp"):
                k = getattr(self, "_fk", 0)
                self._fk = (k + 1) % 6
                self.e(f"v_exp_f32 {vr(TMP(0, 0) + k)}, {vr(MC(0))}")

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
    def iteration(self, i):
        """tile j = %[j], j % 6 == i: S(j) in buffer P = i & 1; P(j-1) and the destination of S(j+1) in buffer P^1; the
        V tile of phase Y in slot (i-1) % 3, the K tile of phase X in slot (i+1) % 3; K(j+2) goes to slot (i+2) % 3, V(j+1)
        to slot (i+1) % 3.  On entry the first PF V fragments are on their way and fragment 0 has landed."""
        P = i & 1
        vs, ks_, vs_next = (i - 1) % 3, (i + 1) % 3, i % 3
        self.lds_log = [("V", 0, f) for f in range(PF) for _ in range(2)]
        self.e(f"; ---- iteration, tile j %% 6 == {i}")
        self.e(f"1{i}:")
        if "bar" not in self.ablate:
            self.e("s_waitcnt vmcnt(0)")
            self.e("s_barrier")
        # ---- phase Y: O^T += V^T.P^T of tile j-1 (slot t: chain t & 1, V fragment t >> 1 = 4*s16 + d).  The wait for a
        # fragment sits behind the MFMA of the slot before its first use (that slot's fillers separate it from the
        # consumer).  Fragment f + PF is read into the ring entry fragment f - 1 has left: its low half behind the first
        # MFMA of fragment f, its high half behind the second.  Eight DMA pieces, behind every fourth MFMA.
        for t in range(32):
            ch, f = t & 1, t >> 1
            self.mfma_y(P ^ 1, t)
            self.filler()
            if ch == 1:
                if f < 15:
                    self.wait_frag("V", 0, f + 1)
                else:
                    self.wait_frag("K", 1, 0)
            dma = ch == 1 and f % 2 == 0 and "dma" not in self.ablate
            if dma:
                pc = f // 2
                if pc < NI:
                    self.e(f"s_add_u32 m0, %[dst0], {((i + 2) % 3) * TILE + pc * NW * 1024}")
                else:
                    self.e(f"s_add_u32 m0, %[dst0], {V_RING + ((i + 1) % 3) * TILE + (pc - NI) * NW * 1024}")
            if t == 0 and "sm" not in self.ablate:  # the sums of tile j-1 (accepted at the end of the last iteration)
                for c2 in range(2):
                    self.e(f"v_add_f32 {vr(L(c2))}, {vr(L(c2))}, {vr(LT(c2))}")
            self.sm_step(P, t)
            if f + PF <= 15:
                self.v_read_half(f + PF, ch, 0, vs)
            elif ch == 0:
                self.k_read(f + PF - 16, 1, ks_)
            if dma:
                ksrd, vsrd = (f"s[{K_SRD}:{K_SRD + 3}]", f"s[{V_SRD}:{V_SRD + 3}]") if self.paged else ("%[ksrd]", "%[vsrd]")
                if pc < NI:
                    self.e(f"buffer_load_dwordx4 {vr(V_KGO)}, {ksrd}, %[koff] offen lds")
                    self.e("s_add_u32 %[koff], %[koff], %[k16]")
                else:
                    self.e(f"buffer_load_dwordx4 {vr(V_VGO)}, {vsrd}, %[voff] offen lds")
                    self.e("s_add_u32 %[voff], %[voff], %[v16]")
        # ---- phase X: S^T = K.Q^T of tile j+1 into the buffer P^1 (slot t: chain t & 1, K fragment t >> 1 = 8*kb + ks)
        # paged: this iteration's pieces are out; the descriptors / offsets of the tiles the NEXT iteration stages (K(j+3),
        # V(j+2)), one step per gap, all ahead of the j increment at t == 30
        pg_steps = (self.page_update(False, 3) + self.page_update(True, 2)) if self.paged else []
        assert len(pg_steps) <= 30
        for t in range(32):
            ch, f = t & 1, t >> 1
            self.mfma_x(P ^ 1, t)
            self.filler()
            if ch == 1:
                if f < 15:
                    self.wait_frag("K", 1, f + 1)
                else:
                    self.wait_frag("V", 2, 0)
            if t < len(pg_steps):
                for ln in pg_steps[t]:
                    self.e(ln)
            self.sm_step(P, 32 + t)
            if f + PF <= 15:
                if ch == 0:
                    self.k_read(f + PF, 1, ks_)
            else:
                self.v_read_half(f + PF - 16, ch, 2, vs_next)
            if t == 30:  # (no scalar instruction behind this one writes SCC before the loop test)
                self.e("s_add_u32 %[j], %[j], 1")
                self.e("s_cmp_lt_i32 %[j], %[jend]")
        self.sm_tail(P)
        assert self.lds_log[-2 * PF:] == [("V", 2, f) for f in range(PF) for _ in range(2)]
        # the test of the tile sums: !(lt0 + lt1 <= limit), NaN included (which chain: P64_CHECK)
        if not self.ablate & {"sm", "smx", "smy"}:
            self.e(f"v_add_f32 {vr(TMP(0, 0))}, {vr(LT(0))}, {vr(LT(1))}")
            self.e(f"v_cmp_nge_f32 vcc, 0x{LIMIT:x}, {vr(TMP(0, 0))}")
            self.e("s_cbranch_vccnz 7f")
        self.e("s_cbranch_scc0 8f")

    def steady(self):
        self.e("; steady-state tile loop of prefill64_kernel (generated by tools/gen_p64_stream.py)")
        self.e("s_mov_b32 %[status], 0")
        self.e("s_cmp_ge_i32 %[j], %[jend]")
        self.e("s_cbranch_scc1 9f")
        self.pads()
        if self.paged:  # the descriptors / offsets of the tiles the first iteration stages: K(j+2), V(j+1)
            for is_v, ahead in ((False, 2), (True, 1)):
                for step in self.page_update(is_v, ahead):
                    for ln in step:
                        self.e(ln)
                self.e(f"s_mov_b32 s{(V_SRD if is_v else K_SRD) + 3}, 0x00020000")
        for i in range(1, 6):
            self.e(f"s_cmp_eq_u32 %[entry], {i}")
            self.e(f"s_cbranch_scc1 2{i}f")
        for i in range(6):  # entries: the first V fragments of phase Y; no sums pending
            self.e(f"2{i}:")
            self.lds_log = []
            for f in range(PF):
                self.v_read(f, 0, (i - 1) % 3)
            for c2 in range(2):
                self.e(f"v_mov_b32 {vr(LT(c2))}, 0")
            self.wait_frag("V", 0, 0, pad=True)
            self.e(f"s_branch 1{i}f")
        for i in range(6):
            self.iteration(i)
        self.e("s_branch 10b")
        self.e("7:")  # tile j's sums failed the test: j was already advanced
        self.e("s_sub_u32 %[j], %[j], 1")
        self.e("s_mov_b32 %[status], 1")
        self.e("s_branch 9f")
        self.e("8:")  # j == jend: the sums of the last tile
        if "sm" not in self.ablate:
            for c2 in range(2):
                self.e(f"v_add_f32 {vr(L(c2))}, {vr(L(c2))}, {vr(LT(c2))}")
        self.e("9:")
        self.e("s_waitcnt lgkmcnt(0)")  # fragments read ahead for an iteration that will not run here
        self.pads()
        return self.out

    # ---- the self-contained phases used outside the steady-state loop; `fill`: extra instructions to spread over the
    # gaps; slot: ring slot as an immediate, or None: byte offset in %[kslot] / %[vslot]
    def phase_x(self, PN, sm, fill=None, slot=None):
        fill = list(fill or [])
        share = [fill[len(fill) * t // 32:len(fill) * (t + 1) // 32] for t in range(32)]
        self.lds_log = []
        self.pads()
        if slot is None:
            for ks in range(8):
                self.e(f"v_add_u32 {vr(KRD(ks))}, %[kslot], {vr(KRD(ks))}")
            slot = 0
            moved = True
        else:
            moved = False
        for f in range(PF):
            self.k_read(f, 0, slot)
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
                self.k_read(f + PF, 0, slot)
        if sm:
            self.sm_tail(PN ^ 1)
        if moved:
            for ks in range(8):
                self.e(f"v_subrev_u32 {vr(KRD(ks))}, %[kslot], {vr(KRD(ks))}")
        self.pads()
        return self.out

    def phase_y(self, PP, sm):
        self.lds_log = []
        self.pads()
        for d in range(4):
            self.e(f"v_add_u32 {vr(VRD(d))}, %[vslot], {vr(VRD(d))}")
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
        for d in range(4):
            self.e(f"v_subrev_u32 {vr(VRD(d))}, %[vslot], {vr(VRD(d))}")
        self.pads()
        return self.out

    def last_block(self, P, masked):
        """a wave's LAST tile j (scores in buffer P) in one piece: phase Y of tile j-1 (V tile in slot %[vslot]) with the
        first half of tile j's softmax in its gaps, the second half behind it (no next tile: nothing to hide under), the
        test of the tile sums (status as P64_CHECK).  masked: the tile's mask (key > row + hi or key >= sk -> -inf;
        %[skm1], %[j64]) applied element by element just ahead of the softmax -- the causal diagonal tile, or the ragged
        last one."""
        self.lds_log = []
        self.pads()
        for d in range(4):
            self.e(f"v_add_u32 {vr(VRD(d))}, %[vslot], {vr(VRD(d))}")
        for f in range(PF):
            self.v_read(f)
        if masked:
            for ch in range(2):
                t0 = TMP(ch, 0)
                self.e(f"v_min_i32 {vr(t0)}, %[skm1], {vr(QHI(ch))}")
                self.e(f"v_subrev_u32 {vr(t0)}, %[j64], {vr(t0)}")
                self.e(f"v_sub_u32 {vr(t0)}, {vr(t0)}, {vr(H4)}")  # keys at tile offsets <= t0 stay
            self.e(f"v_mov_b32 {vr(TMP(0, 1))}, 0x{NEG_INF:x}")
        def mask_ops(u):  # (compare, select) of step u's element
            ch, e = u & 1, u >> 1
            k = 32 * (e >> 4) + ((e & 15) & 3) + 8 * ((e & 15) >> 2)
            x = vr(S_BASE(P, ch) + e)
            return f"v_cmp_gt_i32 vcc, {k}, {vr(TMP(ch, 0))}", f"v_cndmask_b32 {x}, {x}, {vr(TMP(0, 1))}, vcc"

        if masked:
            self.out += mask_ops(0)
        self.wait_frag("V", 0, 0, pad=True)
        for t in range(32):
            ch, f = t & 1, t >> 1
            self.mfma_y(P ^ 1, t)
            if ch == 1 and f < 15:
                self.wait_frag("V", 0, f + 1)
            # the element's mask one gap ahead, interleaved with this gap's fma / exp: no instruction directly behind the
            # one whose result it needs
            nxt = mask_ops(t + 1) if masked and t + 1 < 32 else None
            x = vr(S_BASE(P, ch) + (t >> 1))
            if nxt:
                self.e(nxt[0])
            self.e(f"v_fma_f32 {x}, {x}, %[c], {vr(MC(ch))}")
            if nxt:
                self.e(nxt[1])
            self.e(f"v_exp_f32 {x}, {x}")
            if t >= 2:
                self.sm_sum(P, ch, (t >> 1) - 1)
            if f + PF <= 15:
                self.v_read_half(f + PF, ch)
        for d in range(4):
            self.e(f"v_subrev_u32 {vr(VRD(d))}, %[vslot], {vr(VRD(d))}")
        self.sm_batched(P, 32, 64, masked)
        self.check_block()
        self.pads()
        return self.out

    def sm_batched(self, P, u0, u1, masked=False):
        """softmax steps u0..u1-1 of buffer P (u0 > 0, even) with no MFMAs to ride under: the same instructions as sm_step
        would issue, in batches of eight elements stage by stage -- mask, fma, exp, then the row-sum adds and packs of the
        batch BEFORE -- so that no instruction waits for the result of the one just ahead of it (a dependent VALU pair
        issued back to back costs the wave about twice the issue slot)"""
        B = 8
        batches = [list(range(b0, min(b0 + B, u1))) for b0 in range(u0, u1, B)]

        def sums(us):  # sm_step(u) sums element e-1 = the element of step u-2: after a batch's exps, sum ITS elements
            for u in us:
                self.sm_sum(P, u & 1, u >> 1)

        # (step u0's sm_step would first sum element (u0 >> 1) - 1, whose exp ran in the phase before: do those two now)
        for u in (u0 - 2, u0 - 1):
            self.sm_sum(P, u & 1, u >> 1)
        prev = None
        for us in batches:
            if masked:
                for u in us:
                    ch, e = u & 1, u >> 1
                    k = 32 * (e >> 4) + ((e & 15) & 3) + 8 * ((e & 15) >> 2)
                    x = vr(S_BASE(P, ch) + e)
                    self.e(f"v_cmp_gt_i32 vcc, {k}, {vr(TMP(ch, 0))}")
                    self.e(f"v_cndmask_b32 {x}, {x}, {vr(TMP(0, 1))}, vcc")
            for u in us:
                x = vr(S_BASE(P, u & 1) + (u >> 1))
                self.e(f"v_fma_f32 {x}, {x}, %[c], {vr(MC(u & 1))}")
            for u in us:
                x = vr(S_BASE(P, u & 1) + (u >> 1))
                self.e(f"v_exp_f32 {x}, {x}")
            if prev:
                sums(prev)
            prev = us
        sums(prev)

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
        """phase X of tile 1 (slot %[kslot], buffer 1) with the textbook softmax of tile 0 (buffer 0) in its gaps"""
        return self.phase_x(1, False, fill=self.first_lines(0))

    def check_block(self):
        """status bit c = chain c failed the test of its tile sum; passing chains: l += lt"""
        self.e("s_mov_b32 %[status], 0")
        for ch in range(2):
            self.e(f"v_cmp_nge_f32 vcc, 0x{LIMIT:x}, {vr(LT(ch))}")
            self.e(f"s_cbranch_vccnz {ch + 1}f")
            self.e(f"v_add_f32 {vr(L(ch))}, {vr(L(ch))}, {vr(LT(ch))}")
            self.e(f"s_branch {ch + 3}f")
            self.e(f"{ch + 1}:")
            self.e(f"s_or_b32 %[status], %[status], {1 << ch}")
            self.e(f"{ch + 3}:")
        return self.out

    def redo_block(self, P, ch):
        """%[kslot]: byte offset of the tile's K ring slot"""
        self.pads()
        t2 = TMP(ch, 2)
        for kb in range(2):
            for ks in range(8):
                sn = S_BASE(P, ch) + 16 * kb
                self.e(f"v_add_u32 {vr(t2)}, %[kslot], {vr(KRD(ks))}")
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
        """%[off]: byte offset of the tile's first row (running: left at the next tile's), %[dst]: LDS byte address of this
        wave's first piece in the slot, %[step]: 16 rows in bytes, %[srd]: the K / V descriptor of the (batch, head)"""
        go = V_VGO if is_v else V_KGO
        for p in range(NI):
            self.e(f"s_add_u32 m0, %[dst], {p * NW * 1024}")
            self.e("s_nop 0")
            self.e(f"buffer_load_dwordx4 {vr(go)}, %[srd], %[off] offen lds")
            self.e("s_add_u32 %[off], %[off], %[step]")
        return self.out

    def epilogue_block(self):
        """O / l of both chains -> global memory as whole rows (reference prefill.cuh:600-612): 1/l (1 for a row without
        keys), pack, this wave's LDS staging area (32 rows of 272 bytes, one chain at a time), rows back as 16-byte
        pieces, buffer stores (rows >= seqlen_q fall outside the descriptor and are dropped).
        %[wr] = stage + r*272 + 8h, %[rd] = stage + (lane/16)*272 + 16*(lane%16), %[ovoff] = (first row of the wave +
        lane/16) * row bytes + 16*(lane%16), %[osb4] = 4 rows in bytes"""
        T = lambda i: S_BASE(0, 0) + i  # temporaries: the S buffers are free now
        inv = [T(0), T(1)]
        self.pads()
        for ch in range(2):
            t0, t1 = T(2 + 2 * ch), T(3 + 2 * ch)
            self.e(f"v_mov_b32 {vr(t0)}, {vr(L(ch))}")
            self.e(f"v_mov_b32 {vr(t1)}, {vr(L(ch))}")
        self.e("s_nop 1")
        for ch in range(2):
            self.e(f"v_permlane32_swap_b32 {vr(T(2 + 2 * ch))}, {vr(T(3 + 2 * ch))}")
        self.e("s_nop 1")
        for ch in range(2):
            t0, t1 = T(2 + 2 * ch), T(3 + 2 * ch)
            self.e(f"v_add_f32 {vr(t0)}, {vr(t0)}, {vr(t1)}")
            self.e(f"v_rcp_f32 {vr(t1)}, {vr(t0)}")
            self.e(f"v_cmp_lt_f32 vcc, 0, {vr(t0)}")
            self.e("s_nop 0")
            self.e(f"v_cndmask_b32 {vr(inv[ch])}, 1.0, {vr(t1)}, vcc")
        self.e(f"s_mov_b32 s{S_T0}, 0")
        for ch in range(2):
            if ch == 1:
                self.e("s_waitcnt lgkmcnt(0)")
            n = 0
            for d in range(4):
                for g4 in range(4):
                    x = T(8 + 4 * (n % 8))
                    n += 1
                    for i in range(4):
                        self.e(f"v_accvgpr_read_b32 {vr(x + i)}, {ar(O_BASE(ch, d) + 4 * g4 + i)}")
                    for i in range(4):
                        self.e(f"v_mul_f32 {vr(x + i)}, {vr(x + i)}, {vr(inv[ch])}")
                    self.e(f"{self.cvt} {vr(x)}, {vr(x)}, {vr(x + 1)}")
                    self.e(f"{self.cvt} {vr(x + 1)}, {vr(x + 2)}, {vr(x + 3)}")
                    self.e(f"ds_write_b64 %[wr], {vr(x, 2)} offset:{16 * (4 * d + g4)}")
            self.e("s_waitcnt lgkmcnt(0)")
            R = lambda it: T(48 + 4 * it)
            for it in range(8):
                self.e(f"ds_read_b128 {vr(R(it), 4)}, %[rd] offset:{it * 4 * 272}")
            for it in range(8):
                self.e(f"s_waitcnt lgkmcnt({7 - it})")
                self.e(f"buffer_store_dwordx4 {vr(R(it), 4)}, %[ovoff], %[osrd], s{S_T0} offen" + (" nt" if "o" in NT else ""))
                self.e(f"s_add_u32 s{S_T0}, s{S_T0}, %[osb4]")
        return self.out

    def init_lines(self):
        lines = [f"v_accvgpr_write_b32 {ar(i)}, 0" for i in range(128)]
        lines += [f"v_mov_b32 {vr(r)}, 0" for r in (L(0), L(1), LT(0), LT(1), M(0), M(1), MC(0), MC(1))]
        return lines

    def setup_block(self):
        self.e("s_nop 0")
        return self.out

    def x0_block(self):
        """a new work item: phase X of its tile 0 (slot %[kslot]) into buffer 0; O, l := 0 and the item's mask bounds
        (%[qhi0], %[qhi1]) -> home registers in its gaps.  Q is in its home registers (P64_Q_LDS)."""
        lines = self.init_lines() + [f"v_mov_b32 {vr(QHI(c))}, %[qhi{c}]" for c in range(2)]
        return self.phase_x(0, False, fill=lines)

    def dma_q_block(self):
        """the wave's 64 Q rows of a work item -> its LDS Q buffer, as a K-tile-shaped image (16 pieces of 4 rows; row i at
        256*i, its 16-byte chunks XOR-swizzled by i & 15): whole rows per request, where fragment-shaped loads straight
        from memory (32 rows x 32 bytes per instruction) cost the texture unit four times the cycles.
        %[off]: 0 (running), %[dst]: LDS byte address of the buffer, %[step]: 4 rows in bytes, %[srd]: descriptor of the
        wave's rows (rows >= seqlen_q: zeros)"""
        for p in range(16):
            self.e(f"s_add_u32 m0, %[dst], {p * 1024}")
            self.e("s_nop 0")
            self.e(f"buffer_load_dwordx4 {vr(QGO[p & 3])}, %[srd], %[off] offen" + (" nt" if "q" in NT else "") + " lds")
            self.e("s_add_u32 %[off], %[off], %[step]")
        return self.out

    def q_lds_block(self):
        """Q fragments (B operand of S^T = K.Q^T: row, columns 16*ks + 8h .. +7, as stored; the scale is applied to the
        fp32 scores) from the wave's LDS Q buffer -> home registers.  The image is K-tile-shaped, so the K read addresses
        serve: %[qoff] = the buffer's offset from the K ring's slot 0.  Run between two work items (S buffers free)."""
        T = lambda i: S_BASE(0, 0) + i
        for ks in range(8):
            self.e(f"v_add_u32 {vr(T(ks))}, %[qoff], {vr(KRD(ks))}")
        for c in range(2):
            for ks in range(8):
                self.e(f"ds_read_b128 {ar(Q_BASE(c, ks), 4)}, {vr(T(ks))} offset:{c * 32 * 256}")
        self.e("s_waitcnt lgkmcnt(0)")
        return self.out


ABLATIONS = [("dma",), ("sm",), ("lds",), ("dma", "sm", "lds"), ("dma", "lds"), ("wait",), ("cvt",), ("dma", "sm")]


# ---- operand / clobber lists (C++ names of mfa_prefill64.hip) -------------------------------------------------------------
S_TMP = [f"s{i}" for i in range(84, 86)]
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
    if n == 1:
        return f'"{mode}{{{kind}{lo}}}"'
    return f'"{mode}{{{kind}[{lo}:{lo + n - 1}]}}"'


def emit_block(fh, name, lines_of, outs, ins, exclude=(), extra_clobbers=()):
    n = 0
    for suffix, f16 in (("F16", True), ("BF16", False)):
        lines = lines_of(Stream(f16))
        fh.write(f"#define {name}_{suffix} \\\n")
        for ln in lines:
            fh.write(f'    "{ln}\\n\\t" \\\n')
        fh.write('    ""\n')
        n = sum(1 for ln in lines if not ln.startswith(';') and not ln.endswith(':'))
    fh.write(f"// {name}: {n} instructions\n")
    fh.write(f"#define {name}_OPS : " + ", ".join(outs) + " : " + ", ".join(ins) + " : " + clob(all_regs(exclude) + list(extra_clobbers)) + "\n\n")


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if "--help" in sys.argv or "-h" in sys.argv:
        print(__doc__)
        return
    path = os.path.join(root, "mini-flash-attention_amd", "csrc", "mfa_prefill64_stream.inc")
    if "--out" in sys.argv:  # (tests/test_p64_stream_cpu.py regenerates into a temporary file and compares)
        path = sys.argv[sys.argv.index("--out") + 1]
    KS, VS = '[kslot] "s"(kslot)', '[vslot] "s"(vslot)'
    steady_outs = ['[j] "+s"(j)', '[koff] "+s"(k_off)', '[voff] "+s"(v_off)', '[status] "=&s"(status)']
    steady_ins = ['[jend] "s"(jend)', '[entry] "s"(entry)', '[ksrd] "s"(k_srd)', '[vsrd] "s"(v_srd)', '[k16] "s"(k_step)',
                  '[v16] "s"(v_step)', '[dst0] "s"(dma_dst0)', C_OP]
    with open(path, "w") as fh:
        fh.write("// GENERATED by tools/gen_p64_stream.py -- do not edit.  The instruction streams of prefill64_kernel as inline-asm\n")
        fh.write("// blocks (one text per element type) with their operand lists; register map in the generator's docstring.\n")
        fh.write("// Use:  asm volatile(NAME_F16 NAME_OPS);  inside prefill64_kernel (the operand names are its variables).\n\n")
        # operands -> home registers (pinned inputs; everything else of the reserved range is zeroed or clobbered)
        fixed_ins = [pin("v", KRD(0), 8) + "(k_rd)", pin("v", VRD(0), 4) + "(v_rd)", pin("v", V_KGO, 2) + "(dma_go)",
                     pin("v", QGO[0], 2) + "(q_go01)", pin("v", QGO[2], 2) + "(q_go23)", pin("v", H4, 1) + "(h4)"]
        fixed_regs = (rng("v", KRD(0), 8) + rng("v", VRD(0), 4) + rng("v", V_KGO, 2) + rng("v", QGO[0], 2) +
                      rng("v", QGO[2], 2) + rng("v", H4, 1))
        emit_block(fh, "P64_SETUP", lambda st: st.setup_block(), [], fixed_ins, exclude=fixed_regs)
        emit_block(fh, "P64_DMA_Q", lambda st: st.dma_q_block(), ['[off] "+s"(dma_off)'],
                   ['[dst] "s"(dma_dst)', '[step] "s"(dma_step)', '[srd] "s"(dma_srd)'])
        emit_block(fh, "P64_Q_LDS", lambda st: st.q_lds_block(), [], ['[qoff] "s"(q_lds_off)'])
        emit_block(fh, "P64_X0", lambda st: st.x0_block(), [], [KS, '[qhi0] "v"(qhi0)', '[qhi1] "v"(qhi1)'])
        emit_block(fh, "P64_FIRST0", lambda st: st.first_block(0), [], [C_OP])
        emit_block(fh, "P64_X1_FIRST0", lambda st: st.x_first_block(), [], [KS, C_OP])
        emit_block(fh, "P64_STEADY", lambda st: st.steady(), steady_outs, steady_ins)

        def steady_paged(st):
            st.paged = True
            return st.steady()
        emit_block(fh, "P64_STEADY_PG", steady_paged,
                   ['[j] "+s"(j)', '[koff] "=&s"(k_off)', '[voff] "=&s"(v_off)', '[status] "=&s"(status)'],
                   ['[jend] "s"(jend)', '[entry] "s"(entry)', '[k16] "s"(k_step)', '[v16] "s"(v_step)', '[dst0] "s"(dma_dst0)', C_OP,
                    '[tps] "s"(pg_tps)', '[pshift] "s"(pg_shift)', '[tppm1] "s"(pg_tppm1)', '[pagerows] "s"(pg_rows)',
                    '[maxpg] "s"(pg_max)', '[kvrows] "s"(pg_kvrows)', '[kpoollo] "s"(pg_kpool_lo)', '[kpoolhi] "s"(pg_kpool_hi)',
                    '[vpoollo] "s"(pg_vpool_lo)', '[vpoolhi] "s"(pg_vpool_hi)', '[blk] "s"(pg_blk)', '[sb] "s"(k_sb)'],
                   extra_clobbers=[f"s{i}" for i in range(86, 96)])
        emit_block(fh, "P64_SETTAB", lambda st: [f"v_mov_b32 {vr(TAB)}, %[tab]"], [], ['[tab] "v"(tab_cur)'])
        for pn in range(2):
            emit_block(fh, f"P64_X{pn}_SM", lambda st, pn=pn: st.phase_x(pn, True), [], [KS, C_OP])
            emit_block(fh, f"P64_Y{pn}", lambda st, pn=pn: st.phase_y(pn, False), [], [VS])
            emit_block(fh, f"P64_Y{pn}_SM", lambda st, pn=pn: st.phase_y(pn, True), [], [VS, C_OP])
            emit_block(fh, f"P64_LAST{pn}", lambda st, pn=pn: st.last_block(pn, False), ['[status] "=&s"(status)'], [VS, C_OP])
            emit_block(fh, f"P64_LAST{pn}_M", lambda st, pn=pn: st.last_block(pn, True), ['[status] "=&s"(status)'],
                       [VS, C_OP, '[skm1] "s"(skm1)', '[j64] "s"(j64)'])
            emit_block(fh, f"P64_SM2_{pn}", lambda st, pn=pn: st.sm_second_half(pn), [], [C_OP])
            emit_block(fh, f"P64_MASK{pn}", lambda st, pn=pn: st.mask_block(pn), [], ['[skm1] "s"(skm1)', '[j64] "s"(j64)'])
            for ch in range(2):
                emit_block(fh, f"P64_REDO{pn}{ch}", lambda st, pn=pn, ch=ch: st.redo_block(pn, ch), [],
                           ['[skm1] "s"(skm1)', '[j64] "s"(j64)', KS, C_OP])
        emit_block(fh, "P64_CHECK", lambda st: st.check_block(), ['[status] "=&s"(status)'], [])
        for nm, isv in (("P64_DMA_K", False), ("P64_DMA_V", True)):
            emit_block(fh, nm, lambda st, isv=isv: st.dma_block(isv), ['[off] "+s"(dma_off)'],
                       ['[dst] "s"(dma_dst)', '[step] "s"(dma_step)', '[srd] "s"(dma_srd)'])
        emit_block(fh, "P64_EPILOGUE", lambda st: st.epilogue_block(), [],
                   ['[wr] "v"(stage_wr)', '[rd] "v"(stage_rd)', '[ovoff] "v"(o_voff)', '[osrd] "s"(o_srd)', '[osb4] "s"(o_step)'],
                   exclude=rng("v", L(0), 2) + rng("v", M(0), 2))
        # home registers -> operands: an empty statement whose outputs are pinned to the homes
        fh.write('#define P64_FINAL_F16 ""\n#define P64_FINAL_BF16 ""\n')
        fh.write("#define P64_FINAL_OPS : " + ", ".join([pin("v", L(0), 2, "=") + "(l2)", pin("v", M(0), 2, "=") + "(m2)"]) +
                 " : : \"memory\"\n")
    print("wrote", os.path.relpath(path, root) if "--out" not in sys.argv else path)
    if "--dev" in sys.argv:
        # timing-only variants of the steady loop (results are wrong): developer builds (-DMFA_DEV_P64), MFA_P64_DEBUG >> 2 = 1 + index
        dev = os.path.join(root, "mini-flash-attention_amd", "csrc", "mfa_prefill64_stream_dev.inc")
        with open(dev, "w") as fh:
            fh.write("// GENERATED by tools/gen_p64_stream.py --dev -- developer builds only (csrc/mfa_dev.h), not committed.\n\n")
            for n, tags in enumerate(ABLATIONS):
                def ablated(st, tags=tags):
                    st.ablate = set(tags)
                    return st.steady()
                emit_block(fh, f"P64_STEADY_ABL{n + 1}", ablated, steady_outs, steady_ins)
        print("wrote", os.path.relpath(dev, root))


if __name__ == "__main__":
    main()
