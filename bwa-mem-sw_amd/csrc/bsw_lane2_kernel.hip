/*
 * bsw_lane2_kernel.hip — gfx950 kernel: TWO EXTENSIONS PER LANE (inter-task SIMD with packed 16-bit math).
 *
 * The throughput path for the bins bwa's defaults produce: variant H, bwa-style matrix, symmetric gap penalties,
 * h0 + qlen*a + b <= 255, qlen <= 135 (150 bp reads).  A wavefront walks DP row i of 128 seeds together: lane l holds
 * seed A = order[128w + l] in the low and seed B = order[128w + 64 + l] in the high 16 bits of every register, and
 * every max / saturating-subtract / multiply-add of the recurrence (sw_pe_array_sw_extend.v:1797-1816,1863-1866)
 * is one v_pk_*_u16 instruction for both.  The per-lane arithmetic lives in bsw_lane2_core.h (shared with the CPU
 * model of the tests); this file is the wave-level glue: operand staging through LDS, the wave-uniform block
 * dispatch, result records.  As in bsw_lane_kernel.hip only the first band try runs here; bsw_pair_finalize
 * sends seeds that need MAX_BAND_TRY's second pass to the wave-per-task kernel.
 */
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>
#include <stdlib.h>

#include "bsw_device.h"
#ifdef BSW_L2_STAMP
/* profiling build only: per-wave cycle accumulators of the row-loop sections (s_memtime), reported through the
 * result records instead of the alignment results */
__shared__ unsigned long long l2_acc[4][8];
__shared__ unsigned long long l2_last[4];
__device__ void bsw_l2_stamp(int k)
{
    const int wv = threadIdx.x >> 6;
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) { l2_acc[wv][k] += t - l2_last[wv]; l2_last[wv] = t; }
}
#endif
#include "bsw_lane2_core.h"

namespace bsw {

namespace {

template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ int dpp2(int old, int src)
{
    return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, BANK_MASK, false);
}
__device__ __forceinline__ int wave_max2(int x)
{
    x = max(x, dpp2<0x111>(INT_MIN, x));
    x = max(x, dpp2<0x112>(INT_MIN, x));
    x = max(x, dpp2<0x114>(INT_MIN, x));
    x = max(x, dpp2<0x118>(INT_MIN, x));
    x = max(x, dpp2<0x142, 0xa>(INT_MIN, x));
    x = max(x, dpp2<0x143, 0xc>(INT_MIN, x));
    return __builtin_amdgcn_readlane(x, 63);
}
/* wave-wide unsigned MIN of a, MAX of b, MIN of c, the three DPP chains interleaved by hand: every step reads a register written three
 * instructions earlier, so none of the two wait states a DPP read of a fresh VALU result needs is an s_nop (the compiler's
 * version ran the chains one after the other through one register, 6 dependent steps + 5 s_nop 1 each).  Lanes without a
 * source keep their value: the identity of min and max alike.  (s_nop 1 first: the hazard recogniser does not look inside an
 * asm statement, and a, b, c are computed just before it.) */
__device__ __forceinline__ void wave_min_max_min(uint32_t &a, uint32_t &b, uint32_t &c)
{
#define BSW_DPP3U(ctl) "v_min_u32_dpp %[a], %[a], %[a] " ctl "\n\tv_max_u32_dpp %[b], %[b], %[b] " ctl "\n\tv_min_u32_dpp %[c], %[c], %[c] " ctl "\n\t"
    asm volatile("s_nop 1\n\t" BSW_DPP3U("row_shr:1 row_mask:0xf bank_mask:0xf") BSW_DPP3U("row_shr:2 row_mask:0xf bank_mask:0xf")
                 BSW_DPP3U("row_shr:4 row_mask:0xf bank_mask:0xf") BSW_DPP3U("row_shr:8 row_mask:0xf bank_mask:0xf")
                 BSW_DPP3U("row_bcast:15 row_mask:0xa bank_mask:0xf") BSW_DPP3U("row_bcast:31 row_mask:0xc bank_mask:0xf") "s_nop 0"
                 : [a] "+v"(a), [b] "+v"(b), [c] "+v"(c));
#undef BSW_DPP3U
    a = (uint32_t)__builtin_amdgcn_readlane((int)a, 63); b = (uint32_t)__builtin_amdgcn_readlane((int)b, 63); c = (uint32_t)__builtin_amdgcn_readlane((int)c, 63);
}

/* every 4th bit of a 64-bit word (bit `b` of each nibble) gathered into 16 contiguous bits */
__device__ __forceinline__ uint32_t nib_plane(uint64_t w, int b)
{
    uint64_t x = (w >> b) & 0x1111111111111111ull;
    x = (x | (x >> 3)) & 0x0303030303030303ull;
    x = (x | (x >> 6)) & 0x000F000F000F000Full;
    x = (x | (x >> 12)) & 0x000000FF000000FFull;
    x = (x | (x >> 24)) & 0xFFFFull;
    return (uint32_t)x;
}

}  // namespace

#define BSW_L2_TCHUNK 4         /* target words staged per seed in LDS = 64 DP rows */

template <int QB, int WPS, bool VM, bool SYM, bool FUSED>
__global__ __launch_bounds__(256, WPS) void bsw_lane2_kernel(const bsw_dparams P, const int side_arg,
                                                             const uint64_t *__restrict__ seq,
                                                             const bsw_dtask *__restrict__ tasks,
                                                             const uint32_t *__restrict__ order, const uint32_t n,
                                                             bsw_result *__restrict__ out, uint32_t *tail_flag, const bsw_fin fin)
{
    /* *tail_flag counts the workgroups that have a slot (bsw_lane2l_kernel.hip): the next launch of the
     * chunk's chain is released when the count reaches the grid size */
    if (tail_flag && threadIdx.x == 0) __hip_atomic_fetch_add(tail_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    using L = l2::lane2<QB, VM, SYM>;
    constexpr int QMAX = L::QMAX, NW = L::NW, NC = L::NC;
    static_assert(QMAX <= BSW_LANE_QBINS && QMAX <= 256, "row-max key and binning assume at most 256 eh[] columns");
    __shared__ uint64_t lds_t[4][2][BSW_L2_TCHUNK][64];             /* [wave][seed][word][lane] */
    /* per-base match words of both queries: NQ4 quads of four words + NR single words per (wave, seed, base, lane) */
    constexpr int NQ4 = NW / 4, NR = NW % 4;
    __shared__ uint4 lds_m4[4][2][4][NQ4 ? NQ4 : 1][64];
    __shared__ uint32_t lds_m1[4][2][4][NR ? NR : 1][64];
    __shared__ uint32_t lds_wn[4][NC][64];                          /* N planes of both seeds, 16 columns per half */
    __shared__ uint4 lds_k4[NQ4 ? NQ4 : 1][L::KEEP_NONE + 1];                 /* keep-mask table (match_words), same split */
    __shared__ uint32_t lds_k1[NR ? NR : 1][L::KEEP_NONE + 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t w0 = (blockIdx.x * 4u + (uint32_t)wv) * 128u + (uint32_t)lane;
#ifdef BSW_L2_STAMP
    const unsigned long long wl_t0 = __builtin_amdgcn_s_memrealtime();      /* 100 MHz, the same on every CU: the wave's life span */
#endif
    for (int b = (int)threadIdx.x; b <= L::KEEP_NONE; b += 256) {
#pragma unroll
        for (int q4 = 0; q4 < NQ4; ++q4)
            lds_k4[q4][b] = make_uint4(L::keep_word(b, 4 * q4), L::keep_word(b, 4 * q4 + 1), L::keep_word(b, 4 * q4 + 2), L::keep_word(b, 4 * q4 + 3));
#pragma unroll
        for (int r1 = 0; r1 < NR; ++r1) lds_k1[r1][b] = L::keep_word(b, 4 * NQ4 + r1);
    }
    __syncthreads();                                                /* the only barrier: the table is shared by the four waves */

    /* FUSED (a mid-sized chunk of seeds with two sides each, bsw_binparams.fused): the wavefront runs the LEFT sides of its 128
     * seeds and then their RIGHT sides, the score it found as the right side's h0 — a seed's two sides add up to about one read
     * length whatever the split, so the wavefront lives little longer than one launch of the longest side did and the chunk
     * takes one such lifetime instead of two.  Nothing but the two scores is carried from one side to the other: the seeds,
     * the constants and the operands in LDS are set up again (the one-sided instantiation has no loop: its code is what it was). */
    uint32_t lsc2 = 0;                                              /* the left scores of the lane's two seeds (low / high half) */
    /* (one side = one call of this closure: the one-sided instantiation is straight-line code as before — wrapped in a loop of
     * one pass the compiler spilled 336 registers of the 136-column kernel — and the fused one holds two copies, run in turn) */
    const auto run_side = [&](const int side) __attribute__((always_inline)) {
    typename L::state S;
    uint32_t t_off[2], ti[2], nblk = 0, q2[2][NW];
    int ntw[2];
    bool valid[2], has[2];                                          /* has: the seed exists and has this side (a one-sided list holds only such seeds) */
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
            h0 = T.lqlen > 0 ? (FUSED ? (int)((lsc2 >> (16 * x)) & 0xffffu) : out[ti[x]].left.score) : T.h0;          /* h0 = score after the left ext (:1671) */
        }
        has[x] = valid[x] && qlen > 0;
        if (!has[x]) { tlen = 0; qlen = qlen > 0 ? qlen : 1; t_off[x] = 0; }      /* (the target staging reads word t_off + 0 of every lane: a seed without this side has no offset worth reading at) */
        ntw[x] = (tlen + 15) >> 4;
        l2::init_pair(S.p, x, qlen, tlen, h0, min(P.w, wlim));
        uint32_t mb[4][NW];
#pragma unroll
        for (int wd = 0; wd < NW; ++wd) {
            uint32_t p0 = 0, p1 = 0, p2 = 0;
#pragma unroll
            for (int hlf = 0; hlf < 2; ++hlf) {
                const int v = wd * 2 + hlf;
                if (v < (QMAX + 15) / 16) {
                    const uint64_t qw = (has[x] && v * 16 < qlen) ? seq[q_off + v] : 0ull;
                    p0 |= nib_plane(qw, 0) << (hlf * 16);
                    p1 |= nib_plane(qw, 1) << (hlf * 16);
                    p2 |= nib_plane(qw, 2) << (hlf * 16);
                }
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) mb[b][wd] = L::base_match(p0, p1, p2, b);
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
    const auto again = [](int v) { return FUSED ? (int)l2::opaque_s((uint32_t)v) : v; };      /* (an asm statement: the constants are not kept from one side to the other) */
    k.a = again(P.mat[0]); k.pb = again(-P.mat[1]); k.pn = again(-P.mat[24]);
    k.o_del = again(P.o_del); k.e_del = again(P.e_del); k.e_ins = again(P.e_ins); k.oe_ins = again(P.o_ins) + k.e_ins; k.zdrop = again(P.zdrop);
    l2::fill_packed_consts(k);
    L::init_row(S, k);

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

#ifdef BSW_L2_STAMP
    if (lane == 0) { for (int q = 0; q < 8; ++q) l2_acc[wv][q] = 0; l2_last[wv] = __builtin_amdgcn_s_memtime(); }
    const unsigned long long wl_t1 = __builtin_amdgcn_s_memrealtime();      /* the row loop starts */
#endif
#ifdef BSW_L2_SKEW
    /* experiment: the two waves of a SIMD start a launch's first round in the same row phase and stay there */
    if (__builtin_amdgcn_s_getreg(4 | (0 << 6) | (0 << 11)) & 1u) __builtin_amdgcn_s_sleep(BSW_L2_SKEW);
#endif
    for (int i = 0;; ++i) {
        l2::rowp r;
#ifdef BSW_L2_PRIO
        __builtin_amdgcn_s_setprio(BSW_L2_PRIO);
#endif
        l2::row_begin2(S.p, i, r);
        if (__builtin_amdgcn_ballot_w64(r.ACT != 0) == 0) break;
        L2_STAMP(0);
        if (__builtin_expect((i & (BSW_L2_TCHUNK * 16 - 1)) == 0, 0)) {   /* stage the next 128 target bases of every seed (one row in 64: out of line) */
            const int wbase = i >> 4;
            /* all 16 loads are issued before the first is waited for: the index is clamped instead of branched on
             * (words past the end are never used: row i reads word i >> 4 < ntw; seq has slack behind the last word) */
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

        /* wave-uniform column ranges over the active seeds: blocks outside [jlo, jhi] are skipped, blocks that
         * reach past jem (the smallest `end`) run the masked body */
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
        if (__builtin_expect(u.anybite, 0)) {
            const bool bt0 = (r.BITE & 0xffffu) != 0, bt1 = (r.BITE >> 16) != 0;
            u.zl = -wave_max2(max(bt0 ? -l2::half_of(r.ZLO, 0) : INT_MIN, bt1 ? -l2::half_of(r.ZLO, 1) : INT_MIN));
            u.zh = wave_max2(max(bt0 ? l2::half_of(r.ZHI, 0) : INT_MIN, bt1 ? l2::half_of(r.ZHI, 1) : INT_MIN));
        }
        u.nblk = nblk;
        L2_STAMP(1);
        L::row_body(S, k, i, r, u, tb, qp, kp, wn);
        L2_STAMP(4);
    }

    if (FUSED && side == 0) {
        const int s0 = has[0] ? l2::pair_result(S.p, 0).mx : 0, s1 = has[1] ? l2::pair_result(S.p, 1).mx : 0;
        lsc2 = (uint32_t)s0 | ((uint32_t)s1 << 16);                 /* (8-bit scores) */
    }
    l2::sfor<2>([&](auto xi) {
        constexpr int x = decltype(xi)::value;
        if (!has[x]) return;
        const l2::ext_out s = l2::pair_result(S.p, x);
        bsw_ext e;
        e.score = s.mx; e.qle = s.max_j + 1; e.tle = s.max_i + 1; e.gtle = s.max_ie + 1;
        e.gscore = s.gscore; e.max_off = s.max_off; e.aw = P.w; e.cells = s.cells;
#ifdef BSW_L2_STAMP
        e.score = (int)(l2_acc[wv][0] >> 4); e.qle = (int)(l2_acc[wv][1] >> 4); e.tle = (int)(l2_acc[wv][2] >> 4);
        e.gtle = (int)(l2_acc[wv][3] >> 4); e.gscore = (int)(wl_t1 - wl_t0);      /* (gscore: the prologue, 10 ns ticks) */
#if BSW_L2_STAMP == 2
        e.gscore = (int)(l2_acc[wv][4] >> 4);                                     /* (tools/l2_stamps.py: the row tails instead) */
#endif
        /* wave log (tools/wave_timeline.py): start / end in 10 ns ticks and where the wave ran (HW_ID: wave, SIMD, CU, SE, XCC) */
        e.max_off = (int)(wl_t0 & 0x7fffffffu); e.aw = (int)(__builtin_amdgcn_s_memrealtime() & 0x7fffffffu);
        e.cells = ((unsigned)__builtin_amdgcn_s_getreg((4 /* HW_REG_HW_ID */) | (0 << 6) | (15 << 11)) & 0xffffu) |
                  ((unsigned)__builtin_amdgcn_s_getreg((20 /* HW_REG_XCC_ID */) | (0 << 6) | (3 << 11)) << 16);
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
    };
    if constexpr (FUSED) { run_side(0); run_side(1); }
    else run_side(side_arg);
}

/* what the packed formulation needs from the scoring parameters (everything else takes bsw_lane_kernel): a bwa-style
 * matrix and penalties that fit the 8 score bits.  Both recurrence variants and separate deletion / insertion penalties
 * (the RTL's four-penalty datapath, sw_pe_array_proc_element.v:816-819) run here. */
bool lane2_params_ok(const bsw_dparams &P, int variant)
{
    static const bool off = getenv("BSW_NO_LANE2") != nullptr;
    if (off || (variant != BSW_VARIANT_H && variant != BSW_VARIANT_M)) return false;
    const int a = P.mat[0], pb = -P.mat[1], pn = -P.mat[24];
    return a > 0 && pb >= 0 && pn >= 0 && pb >= pn && a + pb < 256 && P.o_del + P.e_del < 256 && P.o_ins + P.e_ins < 256;
}

hipError_t launch_lane2(int qb, const bsw_dparams &P, int variant, int side, const uint64_t *seq, const bsw_dtask *tasks, const uint32_t *order,
                        uint32_t n, bsw_result *out, hipStream_t s, uint32_t *tail_flag, uint32_t *tail_target, const bsw_fin *finp)
{
    bsw_fin fin;
    if (finp) fin = *finp; else { fin.redo = fin.redo_cnt = nullptr; fin.pairs = nullptr; fin.on = 0; fin.group = 0; }
    if (n == 0) return tail_flag ? hipMemsetD32Async((hipDeviceptr_t)tail_flag, 1, 1, s) : hipSuccess;
    const bool sym = P.o_del == P.o_ins && P.e_del == P.e_ins, vm = variant == BSW_VARIANT_M;
    const dim3 grid((n + 511u) / 512u), block(256);
    if (tail_flag && tail_target) *tail_target = grid.x;              /* the flag's value once every workgroup has started */
#define BSW_L2_GO2(QB, WPS, F)                                                                                                \
    do {                                                                                                                      \
        if (!vm && sym) hipLaunchKernelGGL((bsw_lane2_kernel<QB, WPS, false, true, F>), grid, block, 0, s, P, side, seq, tasks, order, n, out, tail_flag, fin);   \
        else if (!vm) hipLaunchKernelGGL((bsw_lane2_kernel<QB, WPS, false, false, F>), grid, block, 0, s, P, side, seq, tasks, order, n, out, tail_flag, fin);    \
        else if (sym) hipLaunchKernelGGL((bsw_lane2_kernel<QB, WPS, true, true, F>), grid, block, 0, s, P, side, seq, tasks, order, n, out, tail_flag, fin);      \
        else hipLaunchKernelGGL((bsw_lane2_kernel<QB, WPS, true, false, F>), grid, block, 0, s, P, side, seq, tasks, order, n, out, tail_flag, fin);              \
    } while (0)
#define BSW_L2_GO(QB, WPS) BSW_L2_GO2(QB, WPS, false)
    /* 72 columns: 72 row registers + the working set = 168 VGPRs, three waves per SIMD (the spills the compiler takes at
     * that bound sit in the prologue and in the cold target-staging / query-N code, none in the row loop: tests/test_isa_audit.py) */
    if (side == 2) BSW_L2_GO2(17, 2, true);                             /* both sides of every seed (136 columns hold either) */
    else if (qb == 9) BSW_L2_GO(9, 3);
    else BSW_L2_GO(17, 2);
#undef BSW_L2_GO2
#undef BSW_L2_GO
    return hipGetLastError();
}

}  // namespace bsw
