/*
 * bsw_lane2l_kernel.hip — gfx950 kernel: two extensions per lane (as bsw_lane2_kernel.hip), 8-column blocks walked by a
 * RUN-TIME LOOP, eh[] row in the ACCUMULATOR register file.
 *
 * Why: the unrolled kernel's code grows with the class width (356 KB for 232 columns) and at one wave per SIMD — all a
 * 232-register row leaves room for — half of its time is instruction-cache misses (profiles/r3/lane2_wide_*).  gfx950
 * still has the VGPR index mode of gfx9: between s_set_gpr_idx_on and _off, M0[7:0] is added to the register number of
 * the operand slots the mode names, and that includes the AccVGPR operand of v_accvgpr_read / v_accvgpr_write
 * (tools/ubench/gpr_idx.hip).  So the row lives in a[0 .. QMAX+7], this row's match words in the AccVGPRs behind it, and
 * ONE copy of each block body (dense / edge / with query Ns) serves every block: per block eight v_accvgpr_write (the
 * columns just computed), eight v_accvgpr_read (the next block's), two more reads for the match words.  The compiler
 * never allocates AccVGPRs here (the kernel needs < 256 VGPRs; audited in the build: no compiler v_accvgpr_*, no
 * scratch), the reservation statement below makes the kernel descriptor allocate them.
 * One wave per SIMD means nothing hides what the wave itself does not issue, so three things a two-wave kernel gets
 * for free are arranged by hand here (profiles/r3/fetch_alignment.txt): every 64-bit instruction of the block loop
 * starts on the 8-byte grid of the instruction stream (a body shifted by one dword ran 7 % slower: aligned statements,
 * all-64-bit encodings, scalar instructions in pairs); the scalar work of a block is fifteen instructions, eight of
 * them riding in the swap statement's pairs; and a wave none of whose queries holds an N — nine in ten, the bins are
 * keyed by it (bsw_stage_kernel.hip) — runs block loops without any test, the ragged first / last block peeled off.
 * The per-lane arithmetic is lane2l in bsw_lane2_core.h (shared with the CPU model of the tests; the DP cell itself is
 * the unrolled kernel's function).  The reference's PE handles qlen <= 255 in one datapath (sw_pe_array_sw_extend.v:101-102).
 */
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>
#include <stdlib.h>

#include "bsw_device.h"
#ifdef BSW_L2L_STAMP
/* profiling build only (make stampl; tools/l2_stamps.py): per-wave cycle accumulators of the row-loop sections (s_memtime), reported
 * through the result records instead of the alignment results */
#define BSW_L2_STAMP 2
__shared__ unsigned long long l2l_acc[4][8];
__shared__ unsigned long long l2l_last[4];
__device__ void bsw_l2_stamp(int k)
{
    const int wv = threadIdx.x >> 6;
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) { l2l_acc[wv][k] += t - l2l_last[wv]; l2l_last[wv] = t; }
}
#endif
#define BSW_L2_ASM_BODY 1       /* block bodies as hand-scheduled asm (bsw_lane2_body_asm.inc) */
#define BSW_L2_GRID 1           /* ... in their 64-bit-aligned encoding: one wave per SIMD sees every fetch bubble */
#include "bsw_lane2_core.h"

namespace bsw {

namespace {

template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ int dpp2l(int old, int src)
{
    return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, BANK_MASK, false);
}
__device__ __forceinline__ int wave_max2l(int x)
{
    x = max(x, dpp2l<0x111>(INT_MIN, x));
    x = max(x, dpp2l<0x112>(INT_MIN, x));
    x = max(x, dpp2l<0x114>(INT_MIN, x));
    x = max(x, dpp2l<0x118>(INT_MIN, x));
    x = max(x, dpp2l<0x142, 0xa>(INT_MIN, x));
    x = max(x, dpp2l<0x143, 0xc>(INT_MIN, x));
    return __builtin_amdgcn_readlane(x, 63);
}
/* wave-wide unsigned MIN of a, MAX of b, MIN of c at once, the three DPP chains interleaved by hand: every step reads a
 * register written three instructions earlier, so none of the two wait states a DPP read of a fresh VALU result needs is an
 * s_nop (the compiler's version ran the chains one after the other through one register, 6 dependent steps + 5 s_nop 1
 * each).  Lanes without a source keep their value: the identity of min and max alike. */
__device__ __forceinline__ void wave_min_max_min(uint32_t &a, uint32_t &b, uint32_t &c)
{
#define BSW_DPP3U(ctl) "v_min_u32_dpp %[a], %[a], %[a] " ctl "\n\tv_max_u32_dpp %[b], %[b], %[b] " ctl "\n\tv_min_u32_dpp %[c], %[c], %[c] " ctl "\n\t"
    /* (s_nop 1 first: a DPP read of a VGPR needs two wait states behind the VALU write of it, and the hazard recogniser does
     * not look inside an asm statement — a, b, c are computed just before it) */
    asm volatile("s_nop 1\n\t" BSW_DPP3U("row_shr:1 row_mask:0xf bank_mask:0xf") BSW_DPP3U("row_shr:2 row_mask:0xf bank_mask:0xf")
                 BSW_DPP3U("row_shr:4 row_mask:0xf bank_mask:0xf") BSW_DPP3U("row_shr:8 row_mask:0xf bank_mask:0xf")
                 BSW_DPP3U("row_bcast:15 row_mask:0xa bank_mask:0xf") BSW_DPP3U("row_bcast:31 row_mask:0xc bank_mask:0xf") "s_nop 0"
                 : [a] "+v"(a), [b] "+v"(b), [c] "+v"(c));
#undef BSW_DPP3U
    a = (uint32_t)__builtin_amdgcn_readlane((int)a, 63); b = (uint32_t)__builtin_amdgcn_readlane((int)b, 63); c = (uint32_t)__builtin_amdgcn_readlane((int)c, 63);
}

/* every 4th bit of a 64-bit word (bit `b` of each nibble) gathered into 16 contiguous bits */
__device__ __forceinline__ uint32_t nib_plane_l(uint64_t w, int b)
{
    uint64_t x = (w >> b) & 0x1111111111111111ull;
    x = (x | (x >> 3)) & 0x0303030303030303ull;
    x = (x | (x >> 6)) & 0x000F000F000F000Full;
    x = (x | (x >> 12)) & 0x000000FF000000FFull;
    x = (x | (x >> 24)) & 0xFFFFull;
    return (uint32_t)x;
}

/* The eh[] row in AccVGPRs a[0 .. QMAX+7] (the last eight are slack: swap8 of the last block reads past the row), the
 * match words of the current row in a[RM .. RM+2NW).  Register numbers in the strings are RELATIVE: the index mode adds
 * M0[7:0] (= 8b, or the word number) to the slot it is enabled for — 0x1 = SRC0 (the AccVGPR of a read), 0x8 = VDST (the
 * AccVGPR of a write); the VGPR side of each instruction sits in the other slot and is not offset. */
template <int QMAX, int NW>
struct acc_row {
    static constexpr int RM = QMAX + 8;
    __device__ __forceinline__ void load8(int b, uint32_t (&T)[8]) const
    {
        asm volatile(".p2align 3\n\ts_set_gpr_idx_on %8, 0x1\n\ts_nop 0\n\t"
                     "v_accvgpr_read_b32 %0, a0\n\tv_accvgpr_read_b32 %1, a1\n\tv_accvgpr_read_b32 %2, a2\n\tv_accvgpr_read_b32 %3, a3\n\t"
                     "v_accvgpr_read_b32 %4, a4\n\tv_accvgpr_read_b32 %5, a5\n\tv_accvgpr_read_b32 %6, a6\n\tv_accvgpr_read_b32 %7, a7\n\t"
                     "s_set_gpr_idx_off\n\ts_nop 0"
                     : "=v"(T[0]), "=v"(T[1]), "=v"(T[2]), "=v"(T[3]), "=v"(T[4]), "=v"(T[5]), "=v"(T[6]), "=v"(T[7])
                     : "s"(8 * b)
                     : "m0");                      /* s_set_gpr_idx_on overwrites M0 */
    }
    __device__ __forceinline__ void store8(int b, const uint32_t (&T)[8])
    {
        asm volatile(".p2align 3\n\ts_set_gpr_idx_on %8, 0x8\n\ts_nop 0\n\t"
                     "v_accvgpr_write_b32 a0, %0\n\tv_accvgpr_write_b32 a1, %1\n\tv_accvgpr_write_b32 a2, %2\n\tv_accvgpr_write_b32 a3, %3\n\t"
                     "v_accvgpr_write_b32 a4, %4\n\tv_accvgpr_write_b32 a5, %5\n\tv_accvgpr_write_b32 a6, %6\n\tv_accvgpr_write_b32 a7, %7\n\t"
                     "s_set_gpr_idx_off\n\ts_nop 0"
                     :
                     : "v"(T[0]), "v"(T[1]), "v"(T[2]), "v"(T[3]), "v"(T[4]), "v"(T[5]), "v"(T[6]), "v"(T[7]), "s"(8 * b)
                     : "m0");
    }
    /* Block b's columns out, block b+1's columns and match bytes in, under one index-mode window: the index serves the
     * writes as destination offset, then (mode switched) the reads, which name a8..a15, as source offset; the match words
     * sit at another index.  Every 64-bit instruction of the statement starts on the 8-byte grid (tools/isa_align.py):
     * the 32-bit scalar instructions come in pairs, each s_set_gpr_idx_* with one the block needs anyway (the word number,
     * the byte selector of v_perm) — which is also the instruction between the mode change and the first indexed move. */
    __device__ __forceinline__ void swap8w(int b, uint32_t (&T)[8], uint32_t &Wc)
    {
        uint32_t wa, wb, st, sel;
        asm volatile(".p2align 3\n\t"
                     "s_set_gpr_idx_on %[i8], 0x8\n\t"
                     "s_lshr_b32 %[st], %[b1], 2\n\t"
                     "v_accvgpr_write_b32 a0, %0\n\tv_accvgpr_write_b32 a1, %1\n\tv_accvgpr_write_b32 a2, %2\n\tv_accvgpr_write_b32 a3, %3\n\t"
                     "v_accvgpr_write_b32 a4, %4\n\tv_accvgpr_write_b32 a5, %5\n\tv_accvgpr_write_b32 a6, %6\n\tv_accvgpr_write_b32 a7, %7\n\t"
                     "s_set_gpr_idx_mode 0x1\n\t"
                     "s_and_b32 %[sel], %[b1], 3\n\t"
                     "v_accvgpr_read_b32 %0, a8\n\tv_accvgpr_read_b32 %1, a9\n\tv_accvgpr_read_b32 %2, a10\n\tv_accvgpr_read_b32 %3, a11\n\t"
                     "v_accvgpr_read_b32 %4, a12\n\tv_accvgpr_read_b32 %5, a13\n\tv_accvgpr_read_b32 %6, a14\n\tv_accvgpr_read_b32 %7, a15\n\t"
                     "s_set_gpr_idx_idx %[st]\n\t"
                     "s_lshl_b32 %[st], %[sel], 16\n\t"
                     "v_accvgpr_read_b32 %[wa], a[%c[RMA]]\n\tv_accvgpr_read_b32 %[wb], a[%c[RMB]]\n\t"
                     "s_set_gpr_idx_off\n\t"
                     "s_or_b32 %[sel], %[sel], %[st]\n\t"
                     "s_add_i32 %[sel], %[sel], 0x0c040c00\n\t"
                     "v_perm_b32 %[Wc], %[wb], %[wa], %[sel]"
                     : "+v"(T[0]), "+v"(T[1]), "+v"(T[2]), "+v"(T[3]), "+v"(T[4]), "+v"(T[5]), "+v"(T[6]), "+v"(T[7]),
                       [wa] "=&v"(wa), [wb] "=&v"(wb), [Wc] "=v"(Wc), [st] "=&s"(st), [sel] "=&s"(sel)
                     : [i8] "s"(8 * b), [b1] "s"(b + 1), [RMA] "i"(RM), [RMB] "i"(RM + NW)
                     : "scc", "m0");
    }
    __device__ __forceinline__ void load8w(int b, uint32_t (&T)[8], uint32_t &Wc) const
    {
        uint32_t wa, wb, st, sel;
        asm volatile(".p2align 3\n\t"
                     "s_set_gpr_idx_on %[i8], 0x1\n\t"
                     "s_lshr_b32 %[st], %[b0], 2\n\t"
                     "v_accvgpr_read_b32 %0, a0\n\tv_accvgpr_read_b32 %1, a1\n\tv_accvgpr_read_b32 %2, a2\n\tv_accvgpr_read_b32 %3, a3\n\t"
                     "v_accvgpr_read_b32 %4, a4\n\tv_accvgpr_read_b32 %5, a5\n\tv_accvgpr_read_b32 %6, a6\n\tv_accvgpr_read_b32 %7, a7\n\t"
                     "s_set_gpr_idx_idx %[st]\n\t"
                     "s_and_b32 %[sel], %[b0], 3\n\t"
                     "v_accvgpr_read_b32 %[wa], a[%c[RMA]]\n\tv_accvgpr_read_b32 %[wb], a[%c[RMB]]\n\t"
                     "s_set_gpr_idx_off\n\t"
                     "s_lshl_b32 %[st], %[sel], 16\n\t"
                     "s_or_b32 %[sel], %[sel], %[st]\n\t"
                     "s_nop 0\n\t"
                     "s_add_i32 %[sel], %[sel], 0x0c040c00\n\t"
                     "v_perm_b32 %[Wc], %[wb], %[wa], %[sel]"
                     : "=&v"(T[0]), "=&v"(T[1]), "=&v"(T[2]), "=&v"(T[3]), "=&v"(T[4]), "=&v"(T[5]), "=&v"(T[6]), "=&v"(T[7]),
                       [wa] "=&v"(wa), [wb] "=&v"(wb), [Wc] "=v"(Wc), [st] "=&s"(st), [sel] "=&s"(sel)
                     : [i8] "s"(8 * b), [b0] "s"(b), [RMA] "i"(RM), [RMB] "i"(RM + NW)
                     : "scc", "m0");
    }
    template <int WD>
    __device__ __forceinline__ void put_rm_s(uint32_t a, uint32_t b)
    {
        asm volatile("v_accvgpr_write_b32 a[%c2], %0\n\tv_accvgpr_write_b32 a[%c3], %1" : : "v"(a), "v"(b), "i"(RM + WD), "i"(RM + NW + WD));
    }
    __device__ __forceinline__ void put_rm(int wd, uint32_t a, uint32_t b)
    {
        /* wd is a compile-time constant at every call site (sfor): dispatch to the literal register numbers */
        l2::sfor<NW>([&](auto wi) { if (decltype(wi)::value == wd) put_rm_s<decltype(wi)::value>(a, b); });
    }
};

}  // namespace

#define BSW_L2_TCHUNK 4         /* target words staged per seed in LDS = 64 DP rows */

template <int QB, int WPS, bool VM, bool SYM>
__global__ __launch_bounds__(256, WPS) void bsw_lane2l_kernel(const bsw_dparams P, const int side,
                                                              const uint64_t *__restrict__ seq,
                                                              const bsw_dtask *__restrict__ tasks,
                                                              const uint32_t *__restrict__ order, const uint32_t n,
                                                              bsw_result *__restrict__ out, uint32_t *tail_flag, const bsw_fin fin)
{
    /* *tail_flag counts the workgroups that have a slot.  When it reaches gridDim.x every slot that frees up stays free, and
     * whoever waits for that (the next launch of the chunk's chain) may have them.  A count, not "the last
     * block has started": the eight XCDs take their blocks round-robin and run ahead of each other by whole workgroups */
    if (tail_flag && threadIdx.x == 0) __hip_atomic_fetch_add(tail_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    using L = l2::lane2l<QB, VM, SYM>;
    using LB = typename L::B;
    constexpr int QMAX = L::QMAX, NW = L::NW, NC = L::NC;
    static_assert(QMAX <= BSW_LANE_QBINS && QMAX <= 232, "row + slack + match words must fit the 256 AccVGPRs");
    /* the accumulator registers this kernel owns (listing them makes the kernel descriptor allocate them) */
    if constexpr (QMAX + 8 + 2 * NW <= 160) asm volatile("; AccVGPRs a0..a159: eh[] row + match words" : : : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159");
    else asm volatile("; AccVGPRs a0..a255: eh[] row + match words" : : : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255");
    __shared__ uint64_t lds_t[4][2][BSW_L2_TCHUNK][64];             /* [wave][seed][word][lane] */
    constexpr int NQ4 = NW / 4, NR = NW % 4;
    __shared__ uint4 lds_m4[4][2][4][NQ4][64];
    __shared__ uint32_t lds_m1[4][2][4][NR ? NR : 1][64];
    __shared__ uint32_t lds_wn[4][NC][64];                          /* N planes of both seeds, 16 columns per half */
    __shared__ uint4 lds_k4[NQ4][L::KEEP_NONE + 1];                 /* keep-mask table (match_words) */
    __shared__ uint32_t lds_k1[NR ? NR : 1][L::KEEP_NONE + 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#ifdef BSW_L2L_WAVELOG
    const unsigned long long wl_t0 = __builtin_amdgcn_s_memrealtime();      /* (profiling build: tools/wave_timeline.py) */
#endif
    const uint32_t w0 = (blockIdx.x * 4u + (uint32_t)wv) * 128u + (uint32_t)lane;
    for (int b = (int)threadIdx.x; b <= L::KEEP_NONE; b += 256) {
#pragma unroll
        for (int q4 = 0; q4 < NQ4; ++q4)
            lds_k4[q4][b] = make_uint4(LB::keep_word(b, 4 * q4), LB::keep_word(b, 4 * q4 + 1), LB::keep_word(b, 4 * q4 + 2), LB::keep_word(b, 4 * q4 + 3));
#pragma unroll
        for (int r1 = 0; r1 < NR; ++r1) lds_k1[r1][b] = LB::keep_word(b, 4 * NQ4 + r1);
    }
    __syncthreads();                                                /* the only barrier: the table is shared by the four waves */
#ifdef L2L_PAD
    asm volatile(".rept " L2L_PAD "\n\ts_nop 0\n\t.endr");
#endif

    typename L::state S;
    acc_row<QMAX, NW> row;
    uint32_t t_off[2], ti[2], nblk = 0, q2[2][NW];
    int ntw[2];
    bool valid[2];
    l2::sfor<2>([&](auto xi) {
        constexpr int x = decltype(xi)::value;
        const uint32_t slot = w0 + 64u * x;
        const uint32_t oslot = slot < n ? order[slot] : BSW_ORDER_NONE;
        valid[x] = oslot != BSW_ORDER_NONE;                          /* (a list's unused tail: bsw_binparams.nsplit) */
        ti[x] = valid[x] ? oslot : 0u;
        const bsw_dtask T = tasks[ti[x]];
        int qlen, tlen, wlim, h0;
        uint32_t q_off;
        if (side == 0) {
            qlen = T.lqlen; tlen = T.ltlen; wlim = T.wlim_l; q_off = T.lq_off; t_off[x] = T.lt_off; h0 = T.h0;
        } else {
            qlen = T.rqlen; tlen = T.rtlen; wlim = T.wlim_r; q_off = T.rq_off; t_off[x] = T.rt_off;
            h0 = T.lqlen > 0 ? out[ti[x]].left.score : T.h0;          /* h0 = score after the left ext (:1671) */
        }
        if (!valid[x]) { tlen = 0; qlen = 1; q_off = 0; t_off[x] = 0; }      /* (task 0's record stands in: it need not have this side) */
        ntw[x] = (tlen + 15) >> 4;
        l2::init_pair(S.p, x, qlen, tlen, h0, min(P.w, wlim));
        uint32_t mb[4][NW];
        /* all query words are requested before the first is waited for (at one wave per SIMD nothing else hides a round
         * trip to HBM per word); the index is clamped, words past the query are zeroed afterwards */
        constexpr int NV = (QMAX + 15) / 16;
        uint64_t qwv[NV];
        const int lastq = max((qlen + 15) / 16 - 1, 0);
#pragma unroll
        for (int v = 0; v < NV; ++v) qwv[v] = seq[q_off + (uint32_t)min(v, lastq)];
#pragma unroll
        for (int wd = 0; wd < NW; ++wd) {
            uint32_t p0 = 0, p1 = 0, p2 = 0;
#pragma unroll
            for (int hlf = 0; hlf < 2; ++hlf) {
                const int v = wd * 2 + hlf;
                if (v < NV) {
                    const uint64_t qw = (valid[x] && v * 16 < qlen) ? qwv[v] : 0ull;
                    p0 |= nib_plane_l(qw, 0) << (hlf * 16);
                    p1 |= nib_plane_l(qw, 1) << (hlf * 16);
                    p2 |= nib_plane_l(qw, 2) << (hlf * 16);
                }
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) mb[b][wd] = LB::base_match(p0, p1, p2, b);
            q2[x][wd] = p2;
#pragma unroll
            for (int b = 0; b < 4; ++b)
                if (wd * 4 + b < QB && __builtin_amdgcn_ballot_w64(((p2 >> (8 * b)) & 0xffu) != 0) != 0) nblk |= 1u << (wd * 4 + b);
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
#pragma unroll
            for (int q4 = 0; q4 < NQ4; ++q4)
                lds_m4[wv][x][b][q4][lane] = make_uint4(mb[b][4 * q4], mb[b][4 * q4 + 1], mb[b][4 * q4 + 2], mb[b][4 * q4 + 3]);
#pragma unroll
            for (int r1 = 0; r1 < NR; ++r1) lds_m1[wv][x][b][r1][lane] = mb[b][4 * NQ4 + r1];
        }
    });
#pragma unroll
    for (int c = 0; c < NC; ++c)
        lds_wn[wv][c][lane] = ((q2[0][c >> 1] >> (16 * (c & 1))) & 0xffffu) | ((q2[1][c >> 1] >> (16 * (c & 1))) << 16);

    l2::consts k;
    k.a = P.mat[0]; k.pb = -P.mat[1]; k.pn = -P.mat[24];
    k.o_del = P.o_del; k.e_del = P.e_del; k.oe_ins = P.o_ins + P.e_ins; k.e_ins = P.e_ins; k.zdrop = P.zdrop;
    l2::fill_packed_consts(k);
    L::init_row(S, k, row);

    const auto qp = [&](int x, int b, uint32_t (&rm)[NW]) {
#pragma unroll
        for (int q4 = 0; q4 < NQ4; ++q4) {
            const uint4 v = lds_m4[wv][x][b][q4][lane];
            rm[4 * q4] = v.x; rm[4 * q4 + 1] = v.y; rm[4 * q4 + 2] = v.z; rm[4 * q4 + 3] = v.w;
        }
#pragma unroll
        for (int r1 = 0; r1 < NR; ++r1) rm[4 * NQ4 + r1] = lds_m1[wv][x][b][r1][lane];
    };
    const auto kp = [&](int b, uint32_t (&kw)[NW]) {
#pragma unroll
        for (int q4 = 0; q4 < NQ4; ++q4) {
            const uint4 v = lds_k4[q4][b];
            kw[4 * q4] = v.x; kw[4 * q4 + 1] = v.y; kw[4 * q4 + 2] = v.z; kw[4 * q4 + 3] = v.w;
        }
#pragma unroll
        for (int r1 = 0; r1 < NR; ++r1) kw[4 * NQ4 + r1] = lds_k1[r1][b];
    };
    const auto wn = [&](int c) { return lds_wn[wv][c][lane]; };
    uint64_t tw[2] = {0ull, 0ull};
#ifndef BSW_L2L_NO_PREFETCH
    /* ONE ROW AHEAD (round 5).  At one wave per SIMD nothing hides an LDS round trip, and every row started with one: target
     * base -> the query's per-base match words.  Neither depends on the row before it, so row i + 1's target bases are taken
     * and its match words requested while row i is still in front of its cell blocks; row i + 1 finds them in registers
     * (16 + 16 of the 256 architectural VGPRs this kernel has to itself).  Only the keep-table entry, which needs the
     * row's own beg, is still read inside the row. */
    uint32_t rmN[2][NW];
    int tbN[2] = {0, 0};
    const auto fetch_row = [&](const int in) {
        if ((in & (BSW_L2_TCHUNK * 16 - 1)) == 0) {                   /* stage the next 128 target bases of every seed */
            const int wbase = in >> 4;
            uint64_t tv[2][BSW_L2_TCHUNK];
            l2::sfor<2>([&](auto xi) {
                constexpr int x = decltype(xi)::value;
                const int last = max(ntw[x] - 1, 0);
#pragma unroll
                for (int q = 0; q < BSW_L2_TCHUNK; ++q) tv[x][q] = seq[t_off[x] + (uint32_t)min(wbase + q, last)];
            });
            l2::sfor<2>([&](auto xi) {
                constexpr int x = decltype(xi)::value;
#pragma unroll
                for (int q = 0; q < BSW_L2_TCHUNK; ++q) lds_t[wv][x][q][lane] = tv[x][q];
            });
        }
        if ((in & 15) == 0) {
            tw[0] = lds_t[wv][0][(in >> 4) & (BSW_L2_TCHUNK - 1)][lane];
            tw[1] = lds_t[wv][1][(in >> 4) & (BSW_L2_TCHUNK - 1)][lane];
        }
        tbN[0] = (int)((tw[0] >> ((in & 15) * 4)) & 7); tbN[1] = (int)((tw[1] >> ((in & 15) * 4)) & 7);
        qp(0, tbN[0] & 3, rmN[0]);
        qp(1, tbN[1] & 3, rmN[1]);
    };
    fetch_row(0);
#endif

#ifdef BSW_L2L_STAMP
    if (lane == 0) { for (int q = 0; q < 8; ++q) l2l_acc[wv][q] = 0; l2l_last[wv] = __builtin_amdgcn_s_memtime(); }
#endif
    for (int i = 0;; ++i) {
        l2::rowp r;
        l2::row_begin2(S.p, i, r);                                /* K3 band clamp, both seeds at once */
        if (__builtin_amdgcn_ballot_w64(r.ACT != 0) == 0) break;
        L2_STAMP(0);
#ifndef BSW_L2L_NO_PREFETCH
        const int tb[2] = {tbN[0], tbN[1]};
        uint32_t rmC[2][NW];
#pragma unroll
        for (int wd = 0; wd < NW; ++wd) { rmC[0][wd] = rmN[0][wd]; rmC[1][wd] = rmN[1][wd]; }
        fetch_row(i + 1);
        const auto qpc = [&](int x, int, uint32_t (&rm)[NW]) {
#pragma unroll
            for (int wd = 0; wd < NW; ++wd) rm[wd] = rmC[x][wd];
        };
        /* (the keep-table entries requested here too, in front of the wave reductions instead of behind them: 2 170 GCUPS
         * either way on 250 bp reads, gpurun_out/s8 — not kept) */
#else
        if ((i & (BSW_L2_TCHUNK * 16 - 1)) == 0) {                    /* stage the next 128 target bases of every seed */
            const int wbase = i >> 4;
            uint64_t tv[2][BSW_L2_TCHUNK];
            l2::sfor<2>([&](auto xi) {
                constexpr int x = decltype(xi)::value;
                const int last = max(ntw[x] - 1, 0);
#pragma unroll
                for (int q = 0; q < BSW_L2_TCHUNK; ++q) tv[x][q] = seq[t_off[x] + (uint32_t)min(wbase + q, last)];
            });
            l2::sfor<2>([&](auto xi) {
                constexpr int x = decltype(xi)::value;
#pragma unroll
                for (int q = 0; q < BSW_L2_TCHUNK; ++q) lds_t[wv][x][q][lane] = tv[x][q];
            });
        }
        if ((i & 15) == 0) {
            tw[0] = lds_t[wv][0][(i >> 4) & (BSW_L2_TCHUNK - 1)][lane];
            tw[1] = lds_t[wv][1][(i >> 4) & (BSW_L2_TCHUNK - 1)][lane];
        }
        const int tb[2] = {(int)((tw[0] >> ((i & 15) * 4)) & 7), (int)((tw[1] >> ((i & 15) * 4)) & 7)};
        const auto &qpc = qp;
#endif

        l2::uni u;
        {
            /* min beg, max end, min end over the ACTIVE seeds: an inactive half reads 0xffff for the minima, 0 for the maximum */
            const uint32_t nact = ~r.ACT;
            uint32_t ra = l2::min_halves(S.p.BEG | nact), rb = l2::max_halves(S.p.END & r.ACT), rc = l2::min_halves(S.p.END | nact);
            wave_min_max_min(ra, rb, rc);
            u.jlo = (int)ra; u.jhi = (int)rb; u.jem = (int)rc;
        }
        u.anybite = __builtin_amdgcn_ballot_w64(r.BITE != 0) != 0;
        u.zl = 0; u.zh = 0;
        if (u.anybite) {
            const bool bt0 = (r.BITE & 0xffffu) != 0, bt1 = (r.BITE >> 16) != 0;
            u.zl = -wave_max2l(max(bt0 ? -l2::half_of(r.ZLO, 0) : INT_MIN, bt1 ? -l2::half_of(r.ZLO, 1) : INT_MIN));
            u.zh = wave_max2l(max(bt0 ? l2::half_of(r.ZHI, 0) : INT_MIN, bt1 ? l2::half_of(r.ZHI, 1) : INT_MIN));
        }
        u.nblk = nblk;
        L2_STAMP(1);
        L::row_body(S, k, i, r, u, tb, qpc, kp, wn, row);
        L2_STAMP(4);
    }

    l2::sfor<2>([&](auto xi) {
        constexpr int x = decltype(xi)::value;
        if (!valid[x]) return;
        const l2::ext_out s = l2::pair_result(S.p, x);
        bsw_ext e;
        e.score = s.mx; e.qle = s.max_j + 1; e.tle = s.max_i + 1; e.gtle = s.max_ie + 1;
        e.gscore = s.gscore; e.max_off = s.max_off; e.aw = P.w; e.cells = s.cells;
#ifdef BSW_L2L_STAMP
        e.score = (int)(l2l_acc[wv][0] >> 4); e.qle = (int)(l2l_acc[wv][1] >> 4); e.tle = (int)(l2l_acc[wv][2] >> 4);
        e.gtle = (int)(l2l_acc[wv][3] >> 4); e.gscore = (int)(l2l_acc[wv][4] >> 4);
#endif
#ifdef BSW_L2L_WAVELOG
        e.max_off = (int)(wl_t0 & 0x7fffffffu); e.aw = (int)(__builtin_amdgcn_s_memrealtime() & 0x7fffffffu);
        e.cells = ((unsigned)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (15 << 11)) & 0xffffu) | ((unsigned)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) << 16);
        e.gscore = (int)__builtin_popcount(nblk);                           /* blocks with a query N in this wave */
#endif
        if (fin.on) {
            /* the launch that computes a seed's LAST side finishes the seed: clip-vs-extend decision, band-retry test, the
             * whole record (or the redo list) — what bsw_pair_finalize did in a launch of its own (bsw_device.h) */
            const bsw_dtask T = tasks[ti[x]];
            if (side == 1 || T.rqlen == 0) {
                bsw_ext Lx = e;
                if (side == 1 && T.lqlen > 0) Lx = out[ti[x]].left;
                if (fin.pairs) { if (side == 0) out[ti[x]].left = e; else out[ti[x]].right = e; }      /* (pair format: the side records stay in scratch) */
                bsw_pair_decide(P, T, ti[x], Lx, e, out, fin.redo, fin.redo_cnt, fin.pairs);
                return;
            }
        }
        if (side == 0) out[ti[x]].left = e; else out[ti[x]].right = e;
    });
}

/* the 232-column class (250 bp reads) at one wave per SIMD; BSW_LANE2L_NARROW=1 also routes the 136-column class here
 * (experiments: the unrolled kernel is faster there) */
hipError_t launch_lane2l(int qb, const bsw_dparams &P, int variant, int side, const uint64_t *seq, const bsw_dtask *tasks,
                         const uint32_t *order, uint32_t n, bsw_result *out, hipStream_t s, uint32_t *tail_flag, uint32_t *tail_target, const bsw_fin *finp)
{
    bsw_fin fin;
    if (finp) fin = *finp; else { fin.redo = fin.redo_cnt = nullptr; fin.pairs = nullptr; fin.on = 0; fin.group = 0; }
    if (n == 0) return tail_flag ? hipMemsetD32Async((hipDeviceptr_t)tail_flag, 1, 1, s) : hipSuccess;
    const bool sym = P.o_del == P.o_ins && P.e_del == P.e_ins, vm = variant == BSW_VARIANT_M;
    const dim3 grid((n + 511u) / 512u), block(256);
    if (tail_flag && tail_target) *tail_target = grid.x;              /* the flag's value once every workgroup has started */
#define BSW_L2L_GO(QB, WPS)                                                                                                     \
    do {                                                                                                                        \
        if (!vm && sym) hipLaunchKernelGGL((bsw_lane2l_kernel<QB, WPS, false, true>), grid, block, 0, s, P, side, seq, tasks, order, n, out, tail_flag, fin);   \
        else if (!vm) hipLaunchKernelGGL((bsw_lane2l_kernel<QB, WPS, false, false>), grid, block, 0, s, P, side, seq, tasks, order, n, out, tail_flag, fin);    \
        else if (sym) hipLaunchKernelGGL((bsw_lane2l_kernel<QB, WPS, true, true>), grid, block, 0, s, P, side, seq, tasks, order, n, out, tail_flag, fin);      \
        else hipLaunchKernelGGL((bsw_lane2l_kernel<QB, WPS, true, false>), grid, block, 0, s, P, side, seq, tasks, order, n, out, tail_flag, fin);              \
    } while (0)
    if (qb == 17) BSW_L2L_GO(17, 1);
    else BSW_L2L_GO(29, 1);
#undef BSW_L2L_GO
    return hipGetLastError();
}

}  // namespace bsw
