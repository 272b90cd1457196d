// Developer-only scaffolding of prefill_fwd_kernel (mfa_prefill.hip): timing ablations and per-workgroup phase stamps.
// A PRODUCT build sees none of it: every macro below is empty / false unless the library is built with
//     MFA_EXTRA_HIPCC_FLAGS="-DMFA_DEV_ABL_MASK=<bits> [-DMFA_DEV_TIMELINE]" python mini-flash-attention_amd/build.py
// (tools/ab_variants.sh, tools/wg_timeline.py).  Results of an ablated build are wrong by construction.
//   bit 1    no next-tile DMA            bit 2    no end-of-tile barrier     bit 4   exp2 replaced by a move
//   bit 8    no P.V MFMAs                bit 64   no end-of-tile vmcnt(0)    bit 256 no tiles (prologue + epilogue only)
//   bit 512  rescale threshold 0         bit 2048 non-temporal DMA everywhere
#pragma once

#ifdef MFA_DEV_ABL_MASK
#define MFA_DEV_ABL(bit) (((MFA_DEV_ABL_MASK) & (bit)) != 0)
#else
#define MFA_DEV_ABL(bit) false
#endif

#ifdef MFA_DEV_TIMELINE
// phase stamps of every workgroup go to the LSE buffer (8 x i64 per workgroup): entry, loop start, loop end, exit,
// HW_ID, XCC_ID, tiles, marker
#define MFA_DEV_STAMP_DECL long long mfa_dev_stamp[4] = {wall_clock64(), 0, 0, 0}
#define MFA_DEV_STAMP(i) mfa_dev_stamp[i] = wall_clock64()
#define MFA_DEV_TIMELINE_ON true
#define MFA_DEV_STAMP_FLUSH(lse, tid, nt)                                                                              \
    do {                                                                                                               \
        if ((lse) && (tid) == 0) {                                                                                     \
            long long* dbg = (long long*)(lse) + (size_t)blockIdx.x * 8;                                               \
            dbg[0] = mfa_dev_stamp[0]; dbg[1] = mfa_dev_stamp[1]; dbg[2] = mfa_dev_stamp[2]; dbg[3] = wall_clock64();  \
            dbg[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);  /* HW_ID */                                           \
            dbg[5] = __builtin_amdgcn_s_getreg((31 << 11) | 20); /* XCC_ID */                                          \
            dbg[6] = (nt);                                                                                             \
            dbg[7] = 0x5A5A5A5A5A5A5A5ALL;                                                                             \
        }                                                                                                              \
    } while (0)
#else
#define MFA_DEV_STAMP_DECL
#define MFA_DEV_STAMP(i)
#define MFA_DEV_TIMELINE_ON false
#define MFA_DEV_STAMP_FLUSH(lse, tid, nt)
#endif
