/*
 * bsw_lane2g_kernel.hip — gfx950 kernel: a SEED PAIR PER GROUP OF EIGHT LANES (16 seeds per wavefront), packed 16-bit math.
 *
 * The kernel between the general ones and the lane kernels.  bsw_lane2_kernel puts 128 seeds on a wavefront and lets one
 * lane walk a whole row: 13 VALU lane-instructions per cell, but a launch lasts one wave's lifetime — ~1 ms per
 * 131-column side — however few seeds it holds, so batches below ~26 k seeds per side went to bsw_wave_kernel /
 * bsw_quad_kernel (one / four seeds per wavefront, int32, 100 - 130 lane-instructions per cell): 8 k - 64 k PE seeds took
 * 0.6 - 2 ms whichever path ran (profiles/r5/crossover_mixed_bins.json).  The reference's operating point is exactly
 * there: four batches of <= 819 tasks in flight (bwa_mem_sw.v:162-170, sw_pe_array_task_parse.v:1600-1648).
 *
 * Mapping (lane2g in bsw_lane2_core.h, which the CPU model of the tests runs).  A group of eight lanes (half a DPP row) owns
 * seed A = order[16 w + k] in the low and seed B = order[16 w + 8 + k] in the high 16 bits of every register; lane g owns
 * the 8-column blocks j0 = 64 s + 8 g, s = 0 .. NS-1, of both eh[] rows in VGPRs.  Per row and live stripe:
 *   phase A  the block's match byte and its OFFER to the columns right of it (f at its exit with nothing entering)
 *   scan     the offers combined across the group: max-plus prefix with a decay of 8 e_ins per lane, three DPP row_shr steps
 *            (lanes that would read the other group of their DPP row are masked), the carry of the stripe before entering
 *            at lane 0
 *   phase B  lane2's block body, once, with the true f entering the block
 *   phase C  H(i, j0 - 1) from the left neighbour (row_shr:1) into the stored eh[j0], the block's folds
 * then three-step butterflies (quad_perm, quad_perm, row_half_mirror — all inside the group) for the row maximum, the first /
 * last non-zero column and eh[end].h, and lane2's packed row tail, computed redundantly by the eight lanes.
 * No LDS: the match bytes of a lane's own columns (4 bases x NS stripes x 2 seeds) and the current 16 target bases of both
 * seeds live in registers.  As in the other lane kernels only the first band try runs here; the epilogue of a seed's last
 * side takes the pair-level decision (bsw_pair_decide) or pushes the seed onto the redo list.
 */
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>
#include <stdlib.h>

#include "bsw_device.h"
#define BSW_L2_ASM_BODY 1       /* block bodies as list-scheduled asm (bsw_lane2_body_asm.inc), compact encoding */
#include "bsw_lane2_core.h"

namespace bsw {

namespace {

/* DPP moves as asm statements WITH their wait states: the operands come out of inline-asm primitives (v_pk_*), which the
 * compiler's hazard recogniser does not count as VALU writes — a DPP read of a VGPR needs two wait states behind the VALU
 * write of it, and none would be inserted.  bound_ctrl:0 = lanes without a source read 0. */
#define BSW_G_DPP(name, ctl)                                                                                              \
    __device__ __forceinline__ uint32_t name(uint32_t x)                                                                  \
    {                                                                                                                     \
        uint32_t d;                                                                                                       \
        asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 " ctl " row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=v"(d) : "v"(x)); \
        return d;                                                                                                         \
    }
#ifdef BSW_L2G_SHFL      /* (debugging: the same moves through ds_bpermute) */
#undef BSW_G_DPP
#define BSW_G_DPP(name, ctl)
__device__ __forceinline__ uint32_t shfl_from(uint32_t x, int src, bool ok) { const uint32_t y = (uint32_t)__shfl((int)x, src, 64); return ok ? y : 0u; }
__device__ __forceinline__ uint32_t dpp_shr1(uint32_t x) { const int l = threadIdx.x & 63; return shfl_from(x, l - 1, (l & 15) >= 1); }
__device__ __forceinline__ uint32_t dpp_shr2(uint32_t x) { const int l = threadIdx.x & 63; return shfl_from(x, l - 2, (l & 15) >= 2); }
__device__ __forceinline__ uint32_t dpp_shr4(uint32_t x) { const int l = threadIdx.x & 63; return shfl_from(x, l - 4, (l & 15) >= 4); }
__device__ __forceinline__ uint32_t dpp_shl7(uint32_t x) { const int l = threadIdx.x & 63; return shfl_from(x, l + 7, (l & 15) + 7 <= 15); }
__device__ __forceinline__ uint32_t dpp_xor1(uint32_t x) { const int l = threadIdx.x & 63; return shfl_from(x, l ^ 1, true); }
__device__ __forceinline__ uint32_t dpp_xor2(uint32_t x) { const int l = threadIdx.x & 63; return shfl_from(x, l ^ 2, true); }
__device__ __forceinline__ uint32_t dpp_hmir(uint32_t x) { const int l = threadIdx.x & 63; return shfl_from(x, (l & ~7) | (7 - (l & 7)), true); }
#endif
BSW_G_DPP(dpp_shr1, "row_shr:1")
BSW_G_DPP(dpp_shr2, "row_shr:2")
BSW_G_DPP(dpp_shr4, "row_shr:4")
BSW_G_DPP(dpp_shl7, "row_shl:7")
BSW_G_DPP(dpp_xor1, "quad_perm:[1,0,3,2]")
BSW_G_DPP(dpp_xor2, "quad_perm:[2,3,0,1]")
BSW_G_DPP(dpp_hmir, "row_half_mirror")
#undef BSW_G_DPP
/* the value of the lane N places to the left IN THE SAME GROUP of eight, 0 where there is none (lanes 8 .. 8+N-1 of the
 * 16-lane DPP row would read the other group: masked) */
template <int N>
__device__ __forceinline__ uint32_t gshr(uint32_t x, int g)
{
    const uint32_t y = N == 1 ? dpp_shr1(x) : N == 2 ? dpp_shr2(x) : dpp_shr4(x);
    return g >= N ? y : 0u;
}
/* lane 7 of the group, delivered to its lane 0 (row_shl:7; the other lanes get something nobody reads) */
__device__ __forceinline__ uint32_t gfrom7(uint32_t x) { return dpp_shl7(x); }
/* butterflies inside the group: lane ^ 1, lane ^ 2, 7 - lane */
__device__ __forceinline__ uint32_t gmax(uint32_t x)
{
    x = l2::pk_max(x, dpp_xor1(x));
    x = l2::pk_max(x, dpp_xor2(x));
    return l2::pk_max(x, dpp_hmir(x));
}
__device__ __forceinline__ uint32_t gmin(uint32_t x)
{
    x = l2::pk_min(x, dpp_xor1(x));
    x = l2::pk_min(x, dpp_xor2(x));
    return l2::pk_min(x, dpp_hmir(x));
}

/* bit b of each of the 8 nibbles of w gathered into 8 contiguous bits */
__device__ __forceinline__ uint32_t nib_plane8(uint32_t w, int b)
{
    uint32_t x = (w >> b) & 0x11111111u;
    x = (x | (x >> 3)) & 0x03030303u;
    x = (x | (x >> 6)) & 0x000F000Fu;
    x = (x | (x >> 12)) & 0x000000FFu;
    return x;
}

}  // namespace

template <int NS, int WPS, bool VM, bool SYM, bool FUSED>
__global__ __launch_bounds__(256, WPS) void bsw_lane2g_kernel(const bsw_dparams P, const int side_arg,
                                                              const uint64_t *__restrict__ seq,
                                                              const bsw_dtask *__restrict__ tasks,
                                                              const uint32_t *__restrict__ order, const uint32_t n,
                                                              bsw_result *__restrict__ out, const bsw_fin fin)
{
    using L = l2::lane2g<NS, VM, SYM>;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, grp = lane >> 3, g = lane & 7;
    const uint32_t w16s = (blockIdx.x * 4u + (uint32_t)__builtin_amdgcn_readfirstlane(wv)) * 16u;       /* the wavefront's first slot (uniform) */

    int lsc[2] = {0, 0};                                /* (side == 2) the score after the left extension = the right side's h0 */

    /* side == 2 (a mid-sized chunk of seeds with two sides each, bsw_fin.group == 2): the wavefront runs the LEFT sides of its 16
     * seeds and then their RIGHT sides — a seed's two sides add up to about one read length whatever the split, so the wavefront
     * lives about as long as one launch of the longest side did, and the chunk takes one such lifetime instead of two */
    constexpr int side = FUSED ? 2 : 0;
    /* (one side = one call of this closure: straight-line code in the one-sided instantiation, two copies run in turn in the
     * fused one — as a loop over the sides the compiler spilled 26 - 75 registers, some inside the row loop) */
    const auto run_side = [&](const int sd) __attribute__((always_inline)) {
    typename L::state S;
    uint32_t mA[NS], mB[NS], WNs[NS], t_off[2], ti[2];
    int ntw[2];
    bool valid[2], has[2];                              /* has: the seed exists and has this side (a list of one side holds only such seeds) */
    uint32_t nqs = 0;                                   /* bit s: some query of the wavefront has an N in stripe s */
    l2::sfor<2>([&](auto xi) {
        constexpr int x = decltype(xi)::value;
        const uint32_t slot = w16s + (uint32_t)grp + 8u * x;
        const uint32_t oslot = slot < n ? order[slot] : BSW_ORDER_NONE;
        valid[x] = oslot != BSW_ORDER_NONE;                          /* (a list's unused tail: bsw_binparams.nsplit) */
        ti[x] = valid[x] ? oslot : 0u;
        const bsw_dtask T = tasks[ti[x]];
        int qlen, tlen, wlim, h0;
        uint32_t q_off;
        if (sd == 0) {
            qlen = T.lqlen; tlen = T.ltlen; wlim = T.wlim_l; q_off = T.lq_off; t_off[x] = T.lt_off; h0 = T.h0;
        } else {
            qlen = T.rqlen; tlen = T.rtlen; wlim = T.wlim_r; q_off = T.rq_off; t_off[x] = T.rt_off;
            h0 = T.lqlen > 0 ? (side == 2 ? lsc[x] : out[ti[x]].left.score) : T.h0;          /* h0 = score after the left ext (:1671) */
        }
        has[x] = valid[x] && qlen > 0;
        if (!has[x]) { tlen = 0; qlen = qlen > 0 ? qlen : 1; }
        ntw[x] = (tlen + 15) >> 4;
        l2::init_pair(S.p, x, qlen, tlen, h0, min(P.w, wlim));
        /* the lane's own columns of every stripe: 8 nibbles = half a packed word; per base the 8 match bits, and the N bits */
        l2::sfor<NS>([&](auto si) {
            constexpr int s = decltype(si)::value;
            const int wq = 4 * s + (g >> 1);
            const uint64_t qw = (has[x] && 16 * wq < qlen) ? seq[q_off + (uint32_t)wq] : 0ull;
            const uint32_t bits = (g & 1) ? (uint32_t)(qw >> 32) : (uint32_t)qw;
            const uint32_t p0 = nib_plane8(bits, 0), p1 = nib_plane8(bits, 1), p2 = nib_plane8(bits, 2);
            const uint32_t np2 = ~p2 & 0xffu;
            const uint32_t m0 = ~p0 & ~p1 & np2, m1 = p0 & ~p1 & np2, m2 = ~p0 & p1 & np2, m3 = p0 & p1 & np2;
            const uint32_t packed = (m0 & 0xffu) | ((m1 & 0xffu) << 8) | ((m2 & 0xffu) << 16) | ((m3 & 0xffu) << 24);
            if (x == 0) { mA[s] = packed; WNs[s] = p2; } else { mB[s] = packed; WNs[s] |= p2 << 16; }
            if (__builtin_amdgcn_ballot_w64(has[x] && p2 != 0) != 0) nqs |= 1u << s;
        });
    });

    /* (the constants are worked out per side, behind the seeds' set-up as in the one-sided kernel: held across the side loop
     * they cost 60 spilled registers, some of them inside the row loop) */
    l2::consts k;
    const auto again = [](int v) { return FUSED ? (int)l2::opaque_s((uint32_t)v) : v; };      /* (an asm statement: the constants are not kept from one side to the other) */
    k.a = again(P.mat[0]); k.pb = again(-P.mat[1]); k.pn = again(-P.mat[24]);
    k.o_del = again(P.o_del); k.e_del = again(P.e_del); k.e_ins = again(P.e_ins); k.oe_ins = again(P.o_ins) + k.e_ins; k.zdrop = again(P.zdrop);
    l2::fill_packed_consts(k);
    const uint32_t E8 = l2::dup16(min(8 * k.e_ins, 255) << 8), E16 = l2::dup16(min(16 * k.e_ins, 255) << 8), E32 = l2::dup16(min(32 * k.e_ins, 255) << 8);
    L::init_row(S, k, g);

    /* the current and the next 16 target bases of both seeds (every lane of a group loads the same words) */
    uint64_t tw[2], twn[2];
    l2::sfor<2>([&](auto xi) {
        constexpr int x = decltype(xi)::value;
        tw[x] = 0ull;
        twn[x] = ntw[x] > 0 ? seq[t_off[x]] : 0ull;
    });

    for (int i = 0;; ++i) {
        l2::rowp r;
        l2::row_begin2(S.p, i, r);                                /* K3 band clamp, both seeds at once */
        if (__builtin_amdgcn_ballot_w64(r.ACT != 0) == 0) break;
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(r.BITE != 0) != 0, 0)) L::zero_dropped(S, r, g);
        if ((i & 15) == 0) {
            const int wn = (i >> 4) + 1;
            l2::sfor<2>([&](auto xi) {
                constexpr int x = decltype(xi)::value;
                tw[x] = twn[x];
                twn[x] = wn < ntw[x] ? seq[t_off[x] + (uint32_t)wn] : 0ull;
            });
        }
        const int tb[2] = {(int)((tw[0] >> ((i & 15) * 4)) & 7), (int)((tw[1] >> ((i & 15) * 4)) & 7)};
        const typename L::rowk rk = L::row_consts(S.p, k, i, tb);

        uint32_t Hc = rk.h1init, Fc = 0;                          /* what enters lane 0 of the next stripe (read by lane 0 only) */
        uint32_t mk2 = 0, Fnz = 0xffffffffu, Lnz = 0, hfin = 0;
        l2::sfor<NS>([&](auto si_c) {
            constexpr int s = decltype(si_c)::value;
            if (__builtin_amdgcn_ballot_w64(L::needs_stripe(rk, r.ACT, s) != 0) == 0) return;
            const uint32_t J0d = l2::dup16(64 * s + 8 * g);
            typename L::stripe_in si;
            const bool nq = (l2::opaque_s(nqs) >> s) & 1u;        /* (wave-uniform; unlikely: the bins keep the queries with an N apart) */
            uint32_t D;
            if (__builtin_expect(nq, 0)) D = L::template phase_a<true>(S.T[s], rk, k, mA[s], mB[s], WNs[s], J0d, si);
            else D = L::template phase_a<false>(S.T[s], rk, k, mA[s], mB[s], WNs[s], J0d, si);
            /* inclusive max-plus scan of the offers over the group; the carry of the stripe before enters at lane 0 */
            uint32_t x = g == 0 ? l2::pk_max(D, l2::pk_subs_vs(Fc, E8)) : D;
            x = l2::pk_max(x, l2::pk_subs_vs(gshr<1>(x, g), E8));
            x = l2::pk_max(x, l2::pk_subs_vs(gshr<2>(x, g), E16));
            x = l2::pk_max(x, l2::pk_subs_vs(gshr<4>(x, g), E32));
            /* (every cross-lane move is a statement of its own, in front of the select that uses it: inside a `g == 0 ? a : move`
             * the compiler runs the move under the lanes' own EXEC mask, and a DPP read of a DISABLED lane returns 0 — lane 1
             * of every group then saw nothing of lane 0) */
            const uint32_t xs = gshr<1>(x, g);
            uint32_t f = g == 0 ? Fc : xs;                        /* what enters this lane's block */
            uint32_t hl, mkb, nz8;
            if (__builtin_expect(nq, 0)) L::template phase_b<true>(S.T[s], si, rk, k, hl, f, mkb, nz8);
            else L::template phase_b<false>(S.T[s], si, rk, k, hl, f, mkb, nz8);
            const uint32_t hs = gshr<1>(hl, g);
            const uint32_t Hin = g == 0 ? Hc : hs;
            L::phase_c(S.T[s][0], nz8, si, k, Hin);
            L::foldv(mk2, Fnz, Lnz, mkb, nz8, J0d, k);
            hfin = l2::pk_max(hfin, L::hfin_cand(si, hl));
            Hc = gfrom7(hl);
            Fc = gfrom7(f);
        });
#ifdef BSW_L2G_DEBUG
        if (blockIdx.x == 0 && wv == 0 && grp == 0 && (i < 3 || (i >= 70 && i < 75)))
            printf("i=%d g=%d BEG=%08x END=%08x mk2=%08x Lnz=%08x Fnz=%08x hfin=%08x T00=%08x T01=%08x T10=%08x T11=%08x tb=%d,%d mA=%08x,%08x\n", i, g, rk.BEG2, rk.END2, mk2, Lnz, Fnz, hfin,
                   S.T[0][0], S.T[0][1], S.T[1][0], S.T[1][1], tb[0], tb[1], mA[0], mA[1]);
#endif
        mk2 = gmax(mk2); Lnz = gmax(Lnz); Fnz = gmin(Fnz); hfin = gmax(hfin);
        l2::row_tail2<SYM>(S.p, k, i, r.ACT, L::hfin_of(rk, hfin), mk2, Fnz, Lnz);      /* K7, K8 for both seeds at once */
    }

    l2::sfor<2>([&](auto xi) {                                   /* (every lane of the group holds the pair's state) */
        constexpr int x = decltype(xi)::value;
        if (has[x]) lsc[x] = l2::pair_result(S.p, x).mx;
    });
    if (g == 0)                                                   /* one lane of the group writes its two seeds */
    l2::sfor<2>([&](auto xi) {
        constexpr int x = decltype(xi)::value;
        if (!has[x]) return;
        const l2::ext_out s = l2::pair_result(S.p, x);
        bsw_ext e;
        e.score = s.mx; e.qle = s.max_j + 1; e.tle = s.max_i + 1; e.gtle = s.max_ie + 1;
        e.gscore = s.gscore; e.max_off = s.max_off; e.aw = P.w; e.cells = s.cells;
        if (fin.on) {
            /* the launch that computes a seed's LAST side finishes the seed: clip-vs-extend decision, band-retry test, the
             * whole record (or the redo list) — bsw_pair_decide (bsw_device.h), as in the other two-seeds-per-lane kernels */
            const bsw_dtask T = tasks[ti[x]];
            if (sd == 1 || T.rqlen == 0) {
                bsw_ext Lx = e;
                if (sd == 1 && T.lqlen > 0) Lx = out[ti[x]].left;
                if (fin.pairs) { if (sd == 0) out[ti[x]].left = e; else out[ti[x]].right = e; }      /* (pair format: the side records stay in scratch) */
                bsw_pair_decide(P, T, ti[x], Lx, e, out, fin.redo, fin.redo_cnt, fin.pairs);
                return;
            }
        }
        if (sd == 0) out[ti[x]].left = e; else out[ti[x]].right = e;
    });
    };
    if constexpr (FUSED) { run_side(0); run_side(1); }
    else run_side(side_arg);
}

/* cols = eh[] columns of the lane class (<= 192: three stripes, <= 256: four) */
hipError_t launch_lane2g(int cols, const bsw_dparams &P, int variant, int side, const uint64_t *seq, const bsw_dtask *tasks, const uint32_t *order,
                         uint32_t n, bsw_result *out, hipStream_t s, const bsw_fin *finp)
{
    bsw_fin fin;
    if (finp) fin = *finp; else { fin.redo = fin.redo_cnt = nullptr; fin.pairs = nullptr; fin.on = 0; fin.group = 0; }
    if (n == 0) return hipSuccess;
    const bool sym = P.o_del == P.o_ins && P.e_del == P.e_ins, vm = variant == BSW_VARIANT_M;
    const dim3 grid((n + 63u) / 64u), block(256);
#define BSW_L2G_GO2(NS, WPS, F)                                                                                              \
    do {                                                                                                                     \
        if (!vm && sym) hipLaunchKernelGGL((bsw_lane2g_kernel<NS, WPS, false, true, F>), grid, block, 0, s, P, side, seq, tasks, order, n, out, fin);   \
        else if (!vm) hipLaunchKernelGGL((bsw_lane2g_kernel<NS, WPS, false, false, F>), grid, block, 0, s, P, side, seq, tasks, order, n, out, fin);    \
        else if (sym) hipLaunchKernelGGL((bsw_lane2g_kernel<NS, WPS, true, true, F>), grid, block, 0, s, P, side, seq, tasks, order, n, out, fin);      \
        else hipLaunchKernelGGL((bsw_lane2g_kernel<NS, WPS, true, false, F>), grid, block, 0, s, P, side, seq, tasks, order, n, out, fin);              \
    } while (0)
#define BSW_L2G_GO(NS, WPS) do { if (side == 2) BSW_L2G_GO2(NS, WPS, true); else BSW_L2G_GO2(NS, WPS, false); } while (0)
    if (cols <= 192) BSW_L2G_GO(3, 3);
    else BSW_L2G_GO(4, 2);
#undef BSW_L2G_GO2
#undef BSW_L2G_GO
    return hipGetLastError();
}

}  // namespace bsw
