// Developer-only scaffolding of the prefill kernels: timing ablations and per-workgroup phase stamps.
// A PRODUCT build sees none of it: every macro below is empty / false unless the library is built with
//     MFA_EXTRA_HIPCC_FLAGS="-DMFA_DEV_ABL_MASK=<bits> [-DMFA_DEV_TIMELINE] [-DMFA_DEV_P64]" python mini-flash-attention_amd/build.py
// (tools/abl.sh, tools/wg_timeline.py for prefill_fwd_kernel; tools/p64_timeline.py for prefill64_kernel, after
// `python tools/gen_p64_stream.py --dev`).  Results of an ablated build are wrong by construction.
//   bit 1    no next-tile DMA            bit 2    no end-of-tile barrier     bit 4   exp2 replaced by a move
//   bit 8    no P.V MFMAs                bit 64   no end-of-tile vmcnt(0)    bit 256 no tiles (prologue + epilogue only)
//   bit 512  rescale threshold 0         bit 2048 non-temporal DMA everywhere
#pragma once

#ifdef MFA_DEV_ABL_MASK
#define MFA_DEV_ABL(bit) (((MFA_DEV_ABL_MASK) & (bit)) != 0)
#else
#define MFA_DEV_ABL(bit) false
#endif

// non-temporal Q fragment loads / O row stores of prefill_fwd_kernel (A/B switches; the defaults are the measured winners, round 3,
// same box, fp16 B48 H24 D128: O stores non-temporal S=256 causal 0.062 -> 0.051 ms, S=256 non-causal 0.072 -> 0.061, S=128 / 384
// causal -4 %, bf16 varlen B16 S2048 -1 %; Q loads non-temporal +10-20 % on the short shapes: profiles/r03_ab_general_nt.txt)
#ifndef MFA_DEV_NT_Q
#define MFA_DEV_NT_Q false
#endif
#ifndef MFA_DEV_NT_O
#define MFA_DEV_NT_O true
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

// ---- prefill64_kernel (mfa_prefill64.hip): stamps of the workgroup's first two work items by its last wave (16 x u64 per
// workgroup at Sched::dev_ptr, MFA_P64_DEBUG bit 1) and timing-only variants of the loop block (MFA_P64_DEBUG >> 2)
#ifdef MFA_DEV_P64
#define MFA_DEV_P64_ENTRY const unsigned long long mfa_dev_t_entry = __builtin_amdgcn_s_memtime()
#define MFA_DEV_P64_STAMPS(sc, wave, lane, NW)                                                                         \
    unsigned long long* mfa_dev_stamps =                                                                               \
        ((sc).dev_variant & 2) && (sc).dev_ptr && (wave) == (NW) - 1 && (lane) == 0 ? (sc).dev_ptr + (size_t)blockIdx.x * 16 : nullptr; \
    int mfa_dev_items = 0
#define MFA_DEV_P64_FIRST do { if (mfa_dev_stamps) mfa_dev_stamps[0] = mfa_dev_t_entry; } while (0)
#define MFA_DEV_P64_STAMP(i) do { if (mfa_dev_stamps) mfa_dev_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define MFA_DEV_P64_ITEM_DONE(nt, nt_w)                                                                                \
    do {                                                                                                               \
        if (mfa_dev_stamps) mfa_dev_stamps[7] = ((unsigned long long)(nt) << 32) | (unsigned)(nt_w);                   \
        if (++mfa_dev_items == 2) mfa_dev_stamps = nullptr; /* the second item: the steady-state item boundary */       \
        else if (mfa_dev_stamps) mfa_dev_stamps[0] = __builtin_amdgcn_s_memtime();                                     \
    } while (0)
#define MFA_DEV_P64_VARIANT(sc) (__builtin_expect(((sc).dev_variant >> 2) != 0, 0))
#define MFA_DEV_P64_RUN_VARIANT(sc)                                                                                    \
    do {                                                                                                               \
        switch ((sc).dev_variant >> 2) {                                                                               \
        case 1: P64_RUN(P64_STEADY_ABL1); break;                                                                       \
        case 2: P64_RUN(P64_STEADY_ABL2); break;                                                                       \
        case 3: P64_RUN(P64_STEADY_ABL3); break;                                                                       \
        case 4: P64_RUN(P64_STEADY_ABL4); break;                                                                       \
        case 5: P64_RUN(P64_STEADY_ABL5); break;                                                                       \
        case 6: P64_RUN(P64_STEADY_ABL6); break;                                                                       \
        case 7: P64_RUN(P64_STEADY_ABL7); break;                                                                       \
        default: P64_RUN(P64_STEADY_ABL8); break;                                                                      \
        }                                                                                                              \
    } while (0)
#else
#define MFA_DEV_P64_ENTRY
#define MFA_DEV_P64_STAMPS(sc, wave, lane, NW)
#define MFA_DEV_P64_FIRST
#define MFA_DEV_P64_STAMP(i)
#define MFA_DEV_P64_ITEM_DONE(nt, nt_w)
#define MFA_DEV_P64_VARIANT(sc) false
#define MFA_DEV_P64_RUN_VARIANT(sc)
#endif
