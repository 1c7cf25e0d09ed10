/*
 * bsw_quad_kernel.hip — gfx950 kernel: FOUR SEEDS PER WAVEFRONT, one 16-lane DPP row per seed, row-synchronous banded
 * affine-gap extension (ksw_extend2) with the mem_chain2aln driver fused in.  The general path for queries up to 255
 * bases (any 5x5 matrix, int32 scores, band retries in-kernel): small batches, the redo list of the lane kernels, the
 * reference's 819-task wire batches, the scalar ksw_extend2 entry points.
 *
 * Replaces, per seed, what one RTL processing element does
 *   sw_pe_array_proc_element.v:1270-1446  (left ext, right ext, decision, 5-word record)
 *   sw_pe_array_sw_extend.v:1639-1705     (band-retry loop, row loop, II=1 cell pipeline)
 * with CPU (bwa) semantics — RTL quirks Q1-Q7 of SURVEY.md §8a are not reproduced.  The reference's task_parse keeps
 * its 20 PEs busy from any batch of >= 20 tasks (sw_pe_array_task_parse.v:1600-1648); a wavefront per seed
 * (bsw_wave_kernel.hip) keeps most of ITS 64 lanes outside the live band: at qlen 131 a lane owns 3 columns, the band that
 * zero-trimming leaves alive is ~50 columns wide, and every row pays a 6-step scan and two 6-step reductions for one seed.
 *
 * Mapping.  A seed owns a DPP row (16 lanes): every cross-lane step of the recurrence — the F prefix scan (row_shr
 * 1/2/4/8), H(i,j-1) (row_shr:1), the row maximum with arg-max, the first / last non-zero eh[] entry (row_ror 8/4/2/1
 * butterflies), the carries between stripes (row_newbcast:15) — stays inside the row, so four seeds share a wavefront
 * and need no wave-wide scan at all.  Columns are STRIPED, two per lane: lane l owns eh[] entries j = 32 c + 2 l and
 * 32 c + 2 l + 1 of stripe c = 0 .. S-1, in VGPRs for the whole extension (the RTL's 256x16b eh_arr BRAM,
 * sw_pe_array_sw_extend_eh_arr.v).  A row then costs only the stripes between the smallest `beg` and the largest `end` of
 * the wavefront's seeds (a wave-uniform dispatch over statically addressed registers) — work follows the live band, not
 * the query length — and one scan serves 32 columns.
 *   - F(i,j): with A_j = max(base_j - oe_ins, 0) a lane's pair offers D_l = max(A_j0 - e_ins, A_j1) to the columns right
 *     of it; f entering a lane's first column is an exclusive prefix max of B_l = D_l + 2 l e_ins; the value entering the
 *     stripe is carried as Bm1 = f(32c) - 2 e_ins, so f_l = max(Bm1, P_{l-1}) - 2 (l-1) e_ins with one formula for every
 *     lane, and the lane's second column takes max(f - e_ins, A_j0) (valid because o_ins >= 0: f - oe_ins never beats
 *     f - e_ins);
 *   - writes are masked to [beg, end] so stale eh[] entries survive exactly as on the CPU;
 *   - the seeds of a wavefront walk their OWN rows (row index, target base, range are per-lane values, equal inside a
 *     row): a seed that ends a band try or a side re-initialises under its own lanes while the others wait.
 */
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>
#include <utility>

#include "bsw_device.h"

namespace bsw {

namespace {

template <int CTRL>
__device__ __forceinline__ int qdpp(int old, int src)
{
    return __builtin_amdgcn_update_dpp(old, src, CTRL, 0xf, 0xf, false);
}

constexpr int NEGQ = -(1 << 29);                      /* "minus infinity" that survives a few thousand subtractions of e_ins */

template <class F, int... I>
__device__ __forceinline__ void qfor_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void qfor(F &&f) { qfor_impl(f, std::make_integer_sequence<int, N>{}); }

/* four maxima over the 16 lanes of a DPP row at once, the result in every lane of the row (rotations stay inside the row).
 * One statement of four interleaved chains: every DPP read is three instructions behind its producer, so none of the wait
 * states a DPP read of a fresh VALU result needs is an s_nop (from the builtin the compiler makes v_mov_dpp + v_max + a copy
 * of the `old` operand per step: 60 instructions instead of 16). */
__device__ __forceinline__ void row_max4(int &a, int &b, int &c, int &d)
{
#define BSW_Q_ROR(n) "v_max_i32_dpp %0, %0, %0 row_ror:" #n " row_mask:0xf bank_mask:0xf\n\tv_max_i32_dpp %1, %1, %1 row_ror:" #n " row_mask:0xf bank_mask:0xf\n\t" \
                     "v_max_i32_dpp %2, %2, %2 row_ror:" #n " row_mask:0xf bank_mask:0xf\n\tv_max_i32_dpp %3, %3, %3 row_ror:" #n " row_mask:0xf bank_mask:0xf\n\t"
    asm volatile("s_nop 1\n\t" BSW_Q_ROR(8) BSW_Q_ROR(4) BSW_Q_ROR(2) BSW_Q_ROR(1) "s_nop 0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
#undef BSW_Q_ROR
}

/* inclusive prefix max over the 16 lanes of a DPP row: v_max_i32_dpp leaves the lanes without a source (l < shift) as
 * they are, which is the identity here.  (Written out: from the builtin the compiler makes v_mov_dpp + v_max + a
 * re-materialised `old` per step; a DPP read needs two wait states behind the VALU write of its source.) */
__device__ __forceinline__ int row_scan_max(int x)
{
    asm volatile("s_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1"
                 : "+v"(x));
    return x;
}
/* max(p of the lane to the left, b); lane 0 of the row, which has no left neighbour, gets b.  `p` must not have been
 * written by the instruction just before (row_scan_max ends with its own wait states). */
__device__ __forceinline__ int shr1_max(int p, int b)
{
    int d;
    asm volatile("v_mov_b32 %0, %2\n\tv_max_i32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf" : "=&v"(d) : "v"(p), "v"(b));
    return d;
}
/* max(p of the row's lane 15, b) in every lane */
__device__ __forceinline__ int bcast15_max(int p, int b)
{
    int d;
    asm volatile("v_max_i32_dpp %0, %1, %2 row_newbcast:15 row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(p), "v"(b));
    return d;
}
/* a wave-uniform value the compiler must keep in a scalar register and re-read here */
__device__ __forceinline__ uint32_t opaque_su(uint32_t x)
{
    x = (uint32_t)__builtin_amdgcn_readfirstlane((int)x);
    asm volatile("" : "+s"(x));
    return x;
}

struct qside {
    int score, qle, tle, gtle, gscore, max_off, aw;
    unsigned cells;
};

}  // namespace

/* S = stripes of 32 columns a seed may use (cols = 32 S >= qlen + 1) */
/* waves per SIMD the register budget is set for: the row loop itself needs ~95 + 8 S VGPRs; what the (rare, marked unlikely)
 * try / side bookkeeping would like on top of that is spilled there rather than taken from every wave's occupancy */
template <int S>
constexpr int quad_wps() { return S <= 2 ? 4 : S <= 6 ? 3 : 2; }

template <int S, int VAR>
__global__ __launch_bounds__(256, quad_wps<S>()) void bsw_quad_kernel(const bsw_dparams P, const uint64_t *__restrict__ seq,
                                                       const bsw_dtask *__restrict__ tasks,
                                                       const uint32_t *__restrict__ order, const uint32_t n_host,
                                                       const uint32_t *__restrict__ n_dev, uint32_t *__restrict__ next_slot,
                                                       bsw_result *__restrict__ out)
{
    const int l0 = threadIdx.x & 15;
    /* n_dev != NULL: the seed count is produced on the device (redo list of the lane kernels); the grid is then sized by
     * an upper bound and strides over the list */
    const uint32_t n = n_dev ? *n_dev : n_host;
    const int o_del = P.o_del, e_del = P.e_del, o_ins = P.o_ins, e_ins = P.e_ins;
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    const int e2 = 2 * e_ins, e32 = 32 * e_ins;
    const int tries = P.max_band_try > 0 ? P.max_band_try : 1;
    /* the score matrix by query base (K6: sw_pe_array_sw_extend.v:1915-1940): four target bases packed per word + the N row */
    /* (five scalars each, not arrays: an array indexed from inside the stripe lambdas ends up as a lookup table in LDS) */
#define BSW_Q_CP(q) ((uint32_t)(uint8_t)P.mat[q] | ((uint32_t)(uint8_t)P.mat[5 + (q)] << 8) | ((uint32_t)(uint8_t)P.mat[10 + (q)] << 16) | ((uint32_t)(uint8_t)P.mat[15 + (q)] << 24))
    const uint32_t cpl0 = BSW_Q_CP(0), cpl1 = BSW_Q_CP(1), cpl2 = BSW_Q_CP(2), cpl3 = BSW_Q_CP(3), cpl4 = BSW_Q_CP(4);
#undef BSW_Q_CP
    const int cph0 = P.mat[20], cph1 = P.mat[21], cph2 = P.mat[22], cph3 = P.mat[23], cph4 = P.mat[24];

    /* Seeds are handed out one at a time from a device-side counter (*next_slot, zeroed before the launch): a row of lanes
     * that finishes its seed takes the next one while its three neighbours carry on — four seeds of one wavefront need not
     * be equally long (an 819-task wire batch holds sides of 1 to 131 bases), and the grid need not match the list. */
    {
        /* ---- per-seed state (equal in the 16 lanes of a row) ---- */
        uint32_t ti = 0;
        bool alive = true, stop = true;              /* stop / i >= tlen: the current band try has no more rows */
        int side = -2;                               /* -2 no seed yet, -1 seed fetched, 0 left extension, 1 right extension */
        int k = 0, prev = 0, score = 0, aw = P.w, w = 0;
        int qlen = 0, tlen = 0, h0 = 0, wlim = 0, ntw = 0;
        uint32_t q_off = 0, t_off = 0;
        int i = 0, beg = 0, end = 0, mx = 0, max_i = -1, max_j = -1, max_ie = -1, gs = -1, moff = 0;
        unsigned cells = 0;
        uint32_t twl = 0, twh = 0;
        int refill = -1;                             /* the row at which the seed's next 256 target bases must be fetched */
        int Xa[S], Xb[S], Ea[S], Eb[S], pha[S], phb[S];   /* a / b: the lane's first / second column of a stripe */
        uint32_t pla[S], plb[S];
#pragma unroll
        for (int c = 0; c < S; ++c) { Xa[c] = Xb[c] = 0; Ea[c] = Eb[c] = 0; pha[c] = phb[c] = 0; pla[c] = plb[c] = 0; }
        int truesc = 0, qb = 0, rb = 0, qe = 0, re = 0, sc0 = 0, awL = P.w, awR = P.w;

        for (;;) {
            /* ---- seeds whose band try has no more rows: close it; start the next try / the other side / finish ----
             * Written with selects, not branches: the seeds of a wavefront get here at different times, and a divergent region
             * that assigns the row registers makes the compiler keep a shadow copy of every one of them (6 C more VGPRs);
             * the wavefront only comes here when some seed needs it. */
            while (__builtin_expect(__builtin_amdgcn_ballot_w64(alive && (stop || i >= tlen || i == refill)) != 0, 0)) {
                /* (an opaque copy of the lane's column index: what this rare code derives from it per stripe — j, (j-1)*e_ins —
                 * is loop-invariant, and hoisted out of the row loop it costs registers there for nothing) */
                int lq = l0;
                asm volatile("" : "+v"(lq));
                const bool fin = alive && (stop || i >= tlen);
                const bool closing = fin && side >= 0;
                const int sc_new = closing ? mx : score;
                const bool side_done = closing && (sc_new == prev || moff < (aw >> 1) + (aw >> 2) || k + 1 >= tries);   /* P1 (:1837,1859,1822) */
                const bool retry = closing && !side_done;
                const bool recL = side_done && side == 0, recR = side_done && side == 1;
                /* (the seed's task record is re-read here, not kept in registers across the row loop: eleven VGPRs the rows
                 * never look at; the opaque pointer keeps the compiler from hoisting the loads back out) */
                const bsw_dtask *tp = tasks + ti;           /* (ti = 0 before the first seed: a valid record, unused) */
                asm volatile("" : "+v"(tp));
                const bsw_dtask T = *tp;
                {                                     /* K9: the side's record, written once */
                    const int s_qle = max_j + 1, s_tle = max_i + 1, s_gtle = max_ie + 1;
                    if (side_done && lq == 0) {
                        bsw_ext so;
                        so.score = sc_new; so.qle = s_qle; so.tle = s_tle; so.gtle = s_gtle;
                        so.gscore = gs; so.max_off = moff; so.aw = aw; so.cells = cells;
                        if (side == 0) out[ti].left = so; else out[ti].right = so;
                    }
                    awL = recL ? aw : awL;
                    awR = recR ? aw : awR;
                    /* clip or extend to the end (:1672,1674-1675,1666-1667) */
                    const bool localL = gs <= 0 || gs <= sc_new - P.pen_clip5, localR = gs <= 0 || gs <= sc_new - P.pen_clip3;
                    qb = recL ? (localL ? T.qbeg - s_qle : 0) : qb;
                    rb = recL ? (localL ? -s_tle : -s_gtle) : rb;
                    qe = recR ? (localR ? s_qle : (int)T.rqlen) : qe;
                    re = recR ? (localR ? s_tle : s_gtle) : re;
                    truesc = recL ? (localL ? sc_new : gs) : recR ? truesc + (localR ? sc_new : gs) - sc0 : truesc;
                }
                const bool opening = fin && side == -1;                /* the seed's first visit */
                const bool noleft = opening && T.lqlen == 0;
                score = noleft ? T.h0 : (opening ? T.init_score : sc_new);
                truesc = noleft ? T.h0 : truesc;
                sc0 = (recL || noleft) ? score : sc0;                   /* h0 of the right side (:1671) */
                const bool startL = opening && T.lqlen > 0, wantR = recL || noleft;
                const bool startR = wantR && T.rqlen > 0, startS = startL || startR, startT = startS || retry;
                const bool endall = recR || (wantR && T.rqlen == 0);
                if (endall && lq == 0) {              /* the pair-level fields (P2/P3: sw_pe_array_proc_element.v:1662-1669,1684) */
                    bsw_result *r = out + ti;
                    r->tag = T.tag; r->qb = qb; r->qe = qe; r->rb = rb; r->re = re;
                    r->score = score; r->truesc = truesc; r->w = max(awL, awR);
                }
                /* the next seed for the rows that have none (at the start) or have just finished theirs */
                const bool getnext = endall || (fin && side == -2);
                if (__builtin_amdgcn_ballot_w64(getnext) != 0) {
                    uint32_t nxt = 0;
                    if (getnext && lq == 0) nxt = atomicAdd(next_slot, 1u);
                    nxt = (uint32_t)qdpp<0x150>(0, (int)nxt);              /* lane 0 of the row -> the row */
                    const bool has = getnext && nxt < n;
                    if (has) {
                        ti = order[nxt];
                        if (lq < 2) {                 /* a side that does not exist stays neutral in the record */
                            bsw_ext z;
                            z.score = 0; z.qle = z.tle = z.gtle = 0; z.gscore = 0; z.max_off = 0; z.aw = P.w; z.cells = 0;
                            if (lq == 0) out[ti].left = z; else out[ti].right = z;
                        }
                    }
                    alive = getnext ? has : alive;
                    side = getnext ? -1 : side;
                    stop = getnext ? true : stop;
                    truesc = getnext ? 0 : truesc; qb = getnext ? 0 : qb; rb = getnext ? 0 : rb; qe = getnext ? 0 : qe; re = getnext ? 0 : re;
                    sc0 = getnext ? 0 : sc0; awL = getnext ? P.w : awL; awR = getnext ? P.w : awR;
                }
                side = startL ? 0 : startR ? 1 : side;
                qlen = startL ? (int)T.lqlen : startR ? (int)T.rqlen : qlen;
                tlen = startL ? (int)T.ltlen : startR ? (int)T.rtlen : tlen;
                q_off = startL ? T.lq_off : startR ? T.rq_off : q_off;
                t_off = startL ? T.lt_off : startR ? T.rt_off : t_off;
                wlim = startL ? (int)T.wlim_l : startR ? (int)T.wlim_r : wlim;
                h0 = startL ? T.h0 : startR ? sc0 : h0;
                prev = (startS || retry) ? score : prev;
                k = startS ? 0 : retry ? k + 1 : k;
                cells = startS ? 0u : cells;
                ntw = (tlen + 15) >> 4;
                if (__builtin_amdgcn_ballot_w64(startS) != 0) {
                    /* a side starts: its query becomes the per-column score profile.  Lane l fetches packed query word l (16
                     * words = 256 bases cover every class); stripe c's word is then a broadcast from lane c of the row */
                    const uint64_t qw = (startS && 16 * lq < qlen) ? seq[q_off + (uint32_t)lq] : 0ull;
                    const int qwl = (int)(uint32_t)qw, qwh = (int)(uint32_t)(qw >> 32);
                    qfor<S>([&](auto ci) {
                        constexpr int c = decltype(ci)::value;
                        /* columns 32c + 2l, + 1: lanes 0-7 read packed word 2c, lanes 8-15 word 2c + 1; nibbles 2 (l & 7), + 1 */
                        const int j0 = 32 * c + 2 * lq;
                        const uint32_t al = (uint32_t)qdpp<0x150 + 2 * c>(0, qwl), ah = (uint32_t)qdpp<0x150 + 2 * c>(0, qwh);
                        const uint32_t bl = (uint32_t)qdpp<0x150 + 2 * c + 1>(0, qwl), bh = (uint32_t)qdpp<0x150 + 2 * c + 1>(0, qwh);
                        const uint32_t lo32 = lq < 8 ? al : bl, hi32 = lq < 8 ? ah : bh;
                        const uint32_t half = (lq & 4) ? hi32 : lo32;
                        const uint32_t two = half >> ((lq & 3) * 8);
                        const auto prof = [&](int b, int j, uint32_t &lo, int &hi) {
                            b = (j < qlen && b < 4) ? b : 4;
                            /* (masks, not a select chain: the compiler turns the chain into a lookup table in scratch memory) */
                            const uint32_t m0 = 0u - (uint32_t)(b == 0), m1 = 0u - (uint32_t)(b == 1), m2 = 0u - (uint32_t)(b == 2), m3 = 0u - (uint32_t)(b == 3), m4 = 0u - (uint32_t)(b == 4);
                            lo = (cpl0 & m0) | (cpl1 & m1) | (cpl2 & m2) | (cpl3 & m3) | (cpl4 & m4);
                            hi = (int)(((uint32_t)cph0 & m0) | ((uint32_t)cph1 & m1) | ((uint32_t)cph2 & m2) | ((uint32_t)cph3 & m3) | ((uint32_t)cph4 & m4));
                        };
                        uint32_t lo0, lo1;
                        int hi0, hi1;
                        prof((int)(two & 7u), j0, lo0, hi0);
                        prof((int)((two >> 4) & 7u), j0 + 1, lo1, hi1);
                        pla[c] = startS ? lo0 : pla[c]; pha[c] = startS ? hi0 : pha[c];
                        plb[c] = startS ? lo1 : plb[c]; phb[c] = startS ? hi1 : phb[c];
                        __builtin_amdgcn_sched_barrier(0);      /* one stripe at a time: this rare code must not set the kernel's register count */
                    });
                }
                /* a band try starts: K2 first row, closed form (:1979,1957,1974,1818-1821) */
                aw = startT ? P.w << k : aw;
                w = startT ? min(aw, wlim) : w;
                qfor<S>([&](auto ci) {
                    constexpr int c = decltype(ci)::value;
                    const int j0 = 32 * c + 2 * lq;
                    const int x0 = j0 == 0 ? h0 : max(h0 - oe_ins - (j0 - 1) * e_ins, 0);
                    const int x1 = max(h0 - oe_ins - j0 * e_ins, 0);
                    Xa[c] = startT ? x0 : Xa[c]; Xb[c] = startT ? x1 : Xb[c];
                    Ea[c] = startT ? 0 : Ea[c]; Eb[c] = startT ? 0 : Eb[c];
                    __builtin_amdgcn_sched_barrier(0);
                });
                mx = startT ? h0 : mx;
                max_i = startT ? -1 : max_i; max_j = startT ? -1 : max_j; max_ie = startT ? -1 : max_ie;
                gs = startT ? -1 : gs; moff = startT ? 0 : moff; beg = startT ? 0 : beg; end = startT ? qlen : end;
                i = startT ? 0 : i;
                stop = startT ? false : stop;
                /* the next 256 target bases of the seeds that have reached their refill row: lane l holds packed word l */
                refill = startT ? 0 : refill;
                const bool ld = alive && !stop && i < tlen && i == refill;
                if (__builtin_amdgcn_ballot_w64(ld) != 0) {
                    const int wi = (i >> 4) + lq;
                    const uint64_t tv = (ld && wi < ntw) ? seq[t_off + (uint32_t)wi] : 0ull;
                    twl = ld ? (uint32_t)tv : twl;
                    twh = ld ? (uint32_t)(tv >> 32) : twh;
                }
                refill = ld ? refill + 256 : refill;
            }
            if (__builtin_amdgcn_ballot_w64(alive) == 0) break;

            /* ---- one DP row of every live seed ----
             * Every lane runs it: a seed that has finished keeps stepping through garbage that nothing reads (its results are
             * recorded, its lanes fetch nothing from memory) — cheaper than an exec-masked region around the row. */
            /* (the lane's constants are re-derived per row from threadIdx — four cheap instructions — instead of living in
             * registers across the whole kernel: at the register budget of three / four waves per SIMD they were spilled to
             * scratch and RELOADED in every row) */
            int tid = (int)threadIdx.x;
            asm volatile("" : "+v"(tid));
            const int l = tid & 15, l2 = 2 * l, lE2 = l2 * e_ins, lE21 = lE2 - e2;
            /* target base of this row: lane l holds packed word l of the seed's current 256-base chunk */
            const uint32_t tw = (uint32_t)__builtin_amdgcn_ds_bpermute(((tid & 48) | ((i >> 4) & 15)) << 2, (int)((i & 8) ? twh : twl));
            /* K3 band clamp (:1803,1894-1897,1842,1898) */
            beg = max(beg, i - w);
            end = min(min(end, i + w + 1), qlen);
            /* Which stripes this row touches, over the live seeds of the wavefront: stripes cmin .. cmax hold some seed's
             * [beg, end]; stripes imin .. imax lie strictly inside EVERY live seed's range (beg < 32 c, 32 c + 32 <= end) and
             * run the body without masks.  Two packed 16-bit minima / maxima across the four rows, then one bit mask per body. */
            uint32_t m_int, m_edge;
            {
                const uint32_t cb = (uint32_t)(beg >> 5), ce = (uint32_t)max(end, 0) >> 5;
                uint32_t mn = alive ? (cb | (ce << 16)) : 0x001f001fu;     /* min: first stripe | last stripe for the interior */
                uint32_t mxv = alive ? (ce | (cb << 16)) : 0u;              /* max: last stripe | first stripe for the interior */
                uint32_t t;
                asm volatile("s_nop 1\n\t"
                             "v_mov_b32_dpp %2, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                             "v_pk_min_u16 %0, %0, %2\n\t"
                             "v_mov_b32_dpp %2, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                             "v_pk_max_u16 %1, %1, %2\n\t"
                             "s_nop 0\n\t"
                             "v_mov_b32_dpp %2, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                             "v_pk_min_u16 %0, %0, %2\n\t"
                             "v_mov_b32_dpp %2, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                             "v_pk_max_u16 %1, %1, %2"
                             : "+v"(mn), "+v"(mxv), "=&v"(t));
                /* (rows 1 and 3 take the row before them, then rows 2 and 3 take row 1's / row 1+0's: lane 63 holds all four.
                 * The lanes a row_bcast does not write keep `t` from the step before — equal to their own value or stale, and
                 * min / max with a stale copy of an EARLIER partial result of the same reduction changes nothing.) */
                const uint32_t smn = (uint32_t)__builtin_amdgcn_readlane((int)mn, 63), smx = (uint32_t)__builtin_amdgcn_readlane((int)mxv, 63);
                const uint32_t cmin = smn & 0xffffu, imax_p1 = smn >> 16, cmax = smx & 0xffffu, imin_m1 = smx >> 16;
                const uint32_t below = (1u << min(cmin, 31u)) - 1u, upto = (2u << min(cmax, 30u)) - 1u;
                const uint32_t m_run = upto & ~below;                                           /* cmin <= c <= cmax */
                const uint32_t ilo = imin_m1 + 1u;                                              /* imin_m1 < c < imax_p1 */
                const uint32_t ibits = imax_p1 > ilo ? ((1u << min(imax_p1, 31u)) - 1u) & ~((1u << min(ilo, 31u)) - 1u) : 0u;
                m_int = opaque_su(m_run & ibits);
                m_edge = opaque_su(m_run & ~ibits);
            }
            const int tb = (int)((tw >> ((i & 7) * 4)) & 7u);
            const bool tn = tb >= 4;
            const uint32_t sh = (uint32_t)(tb & 3) * 8u;
            /* K4 column 0, CPU semantics (:1795-1796,1835; Q4 avoided) */
            const int h1_init = beg == 0 ? max(h0 - (o_del + __mul24(e_del, i + 1)), 0) : 0;   /* (e_del <= 4096, i < 65536) */
            cells += (unsigned)max(end - beg, 0);
            const int rel = l2 - beg, span_e = end - beg;
            const unsigned span_in = (unsigned)max(span_e, 0), span_wr = (unsigned)max(span_e + 1, 0);

            int Bm1 = NEGQ;                           /* f entering the stripe, minus 2 e_ins; nothing enters the first one */
            int h31 = 0;                              /* H(i, 32c - 1): the previous stripe's last column */
            int bestv = -1, bestk = 0, hl = -1;       /* bestk = 2 c + (second column) */
            uint32_t nzb = 0;                         /* bit 2 c + k: this lane's eh[] entry of stripe c, column k, is non-zero (inside [beg, end]) */
            /* one stripe.  INT: every column of the stripe lies strictly inside every live seed's range — no masks.
             * TNV: some live seed's target base is an N this row (its scores come from the matrix's N row). */
            const auto stripe = [&](auto ci, auto intv, auto tnv) {
                constexpr int c = decltype(ci)::value;
                constexpr bool INT = decltype(intv)::value, TNV = decltype(tnv)::value;
                const int u0 = rel + 32 * c, u1 = u0 + 1;     /* j - beg of the lane's two columns */
                const bool inr0 = INT || (unsigned)u0 < span_in, inr1 = INT || (unsigned)u1 < span_in;
                const bool wr0 = INT || (unsigned)u0 < span_wr, wr1 = INT || (unsigned)u1 < span_wr;
                int s0 = (int)__builtin_amdgcn_sbfe((int)pla[c], sh, 8u), s1 = (int)__builtin_amdgcn_sbfe((int)plb[c], sh, 8u);
                if (TNV) { s0 = tn ? pha[c] : s0; s1 = tn ? phb[c] : s1; }
                int M0 = Xa[c] + s0, M1 = Xb[c] + s1;          /* variant H (:1797) */
                if (VAR == BSW_VARIANT_M) { M0 = Xa[c] ? M0 : 0; M1 = Xb[c] ? M1 : 0; }
                const int ht0 = max(M0, Ea[c]), ht1 = max(M1, Eb[c]);   /* (:1798) */
                int A0 = max((VAR == BSW_VARIANT_M ? M0 : ht0) - oe_ins, 0), A1 = max((VAR == BSW_VARIANT_M ? M1 : ht1) - oe_ins, 0);
                if (!INT) { A0 = inr0 ? A0 : NEGQ; A1 = inr1 ? A1 : NEGQ; }
                /* what the pair offers the columns right of it, scanned over the row's lanes (F recurrence, :1863,1780-1781) */
                const int Pm = row_scan_max(max(A0 - e_ins, A1) + lE2);
                const int f0 = shr1_max(Pm, Bm1) - lE21;
                const int hv0 = max(ht0, f0);         /* (:1809,1944) */
                const int f1 = max(f0 - e_ins, A0);
                const int hv1 = max(ht1, f1);
                Bm1 = bcast15_max(Pm, Bm1) - e32;
                const int en0 = max(max(Ea[c] - e_del, (VAR == BSW_VARIANT_M ? M0 : hv0) - oe_del), 0);   /* (:1866,1770-1771) */
                const int en1 = max(max(Eb[c] - e_del, (VAR == BSW_VARIANT_M ? M1 : hv1) - oe_del), 0);
                /* row maximum, ties -> later j (:1808,1816): columns ascend, so >= keeps the later one */
                const bool take0 = inr0 && hv0 >= bestv;
                bestv = take0 ? hv0 : bestv;
                bestk = take0 ? 2 * c : bestk;
                const bool take1 = inr1 && hv1 >= bestv;
                bestv = take1 ? hv1 : bestv;
                bestk = take1 ? 2 * c + 1 : bestk;
                /* eh[j].h <- H(i,j-1) for j in [beg,end]; eh[end].e <- 0 (:1776,1775): the first column takes the lane to the
                 * left's second one (lane 0: the previous stripe's last), the second column the lane's own first */
                const int hp0 = qdpp<0x111>(h31, hv1);
                h31 = qdpp<0x15F>(0, hv1);
                int xa, xb, ea, eb;
                if (INT) { xa = hp0; xb = hv0; ea = en0; eb = en1; }
                else {
                    const int xn0 = u0 == 0 ? h1_init : hp0, xn1 = u1 == 0 ? h1_init : hv0;
                    const bool jend0 = u0 == span_e, jend1 = u1 == span_e;
                    xa = wr0 ? xn0 : Xa[c];
                    xb = wr1 ? xn1 : Xb[c];
                    ea = inr0 ? en0 : Ea[c];
                    eb = inr1 ? en1 : Eb[c];
                    ea = jend0 ? 0 : ea;
                    eb = jend1 ? 0 : eb;
                    hl = jend0 ? xa : hl;
                    hl = jend1 ? xb : hl;
                }
                Xa[c] = xa; Xb[c] = xb; Ea[c] = ea; Eb[c] = eb;
                const bool nz0 = wr0 && ((xa | ea) != 0), nz1 = wr1 && ((xb | eb) != 0);
                nzb |= (nz0 ? (1u << (2 * c)) : 0u) | (nz1 ? (2u << (2 * c)) : 0u);
            };
            using no_t = std::integral_constant<bool, false>;
            using yes_t = std::integral_constant<bool, true>;
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(tn && alive) != 0, 0)) {
                qfor<S>([&](auto ci) {
                    constexpr int c = decltype(ci)::value;
                    if ((m_int >> c) & 1u) stripe(ci, yes_t{}, yes_t{});
                    if ((m_edge >> c) & 1u) stripe(ci, no_t{}, yes_t{});
                });
            } else {
                qfor<S>([&](auto ci) {
                    constexpr int c = decltype(ci)::value;
                    if ((m_int >> c) & 1u) stripe(ci, yes_t{}, no_t{});
                    if ((m_edge >> c) & 1u) stripe(ci, no_t{}, no_t{});
                });
            }
            /* column of code k = 2 c + (second column): 32 c + 2 l + (k & 1) = 16 k - 15 (k & 1) + 2 l */
            const auto col_of = [&](int kcode) { return 16 * kcode - 15 * (kcode & 1) + l2; };
            int key = bestv < 0 ? -1 : ((bestv << BSW_KEY_BITS) | col_of(bestk));
            int nfz = nzb ? -col_of(__builtin_ctz(nzb)) : INT_MIN;
            int lz = nzb ? col_of(31 - __builtin_clz(nzb)) : -1;
            row_max4(key, nfz, lz, hl);
            const int mrow = key < 0 ? 0 : (key >> BSW_KEY_BITS);
            const int mj = key < 0 ? -1 : (key & ((1 << BSW_KEY_BITS) - 1));
            const int first_nz = nfz == INT_MIN ? INT_MAX : -nfz, last_nz = lz;
            /* eh[end].h; when the range is empty past the band (beg > end) the CPU's h1 is h1_init */
            const int hlast = end < beg ? h1_init : hl;
            /* row tail scalars (K7), by selects: the seeds of a wavefront differ */
            const bool atq = max(beg, end) == qlen;                        /* (:1913,1941,1829-1833) ties -> later i */
            max_ie = (atq && !(gs > hlast)) ? i : max_ie;
            gs = atq ? max(gs, hlast) : gs;
            const bool gt = mrow > mx;                                     /* (:1959,1810,1845,1812-1813) */
            const int di = i - max_i, dj = mj - max_j;                     /* against the OLD maximum */
            const bool zd = P.zdrop > 0 && !gt &&                          /* C ABI only; RTL has no zdrop (Q3) */
                            mx - mrow - __mul24(abs(di - dj), di > dj ? e_del : e_ins) > P.zdrop;
            stop = mrow == 0 || zd;                                        /* (:1942) */
            moff = gt ? max(moff, abs(mj - i)) : moff;
            max_i = gt ? i : max_i;
            max_j = gt ? mj : max_j;
            mx = gt ? mrow : mx;
            /* K8 next-row range, CPU semantics (Q5 avoided) */
            const int nbeg = first_nz < end ? first_nz : end;
            const int last = last_nz >= 0 ? last_nz : nbeg - 1;
            beg = nbeg;
            end = min(last + 2, qlen);
            ++i;
        }

    }
}

template <int S>
static hipError_t launch_qc(int variant, const bsw_dparams &P, const uint64_t *seq, const bsw_dtask *tasks,
                            const uint32_t *order, uint32_t n, const uint32_t *n_dev, uint32_t *next_slot, bsw_result *out, hipStream_t s)
{
    /* one row of lanes per seed up to a grid that fills the machine several times over; beyond that the rows refill
     * themselves from the counter */
    uint32_t blocks = (n + 15u) / 16u;
    if (blocks > 2048u) blocks = 2048u;
    const dim3 grid(blocks), block(256);
    if (variant == BSW_VARIANT_M)
        hipLaunchKernelGGL((bsw_quad_kernel<S, BSW_VARIANT_M>), grid, block, 0, s, P, seq, tasks, order, n, n_dev, next_slot, out);
    else
        hipLaunchKernelGGL((bsw_quad_kernel<S, BSW_VARIANT_H>), grid, block, 0, s, P, seq, tasks, order, n, n_dev, next_slot, out);
    return hipGetLastError();
}

/* cols = eh[] columns of the seed class (qlen + 1 <= cols <= 256); n = seed count (or an upper bound of *n_dev);
 * next_slot = a device word that is ZERO when the kernel starts (the caller's memset on the same stream) */
hipError_t launch_quad(int cols, int variant, const bsw_dparams &P, const uint64_t *seq, const bsw_dtask *tasks,
                       const uint32_t *order, uint32_t n, const uint32_t *n_dev, uint32_t *next_slot, bsw_result *out, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    if (cols <= 64) return launch_qc<2>(variant, P, seq, tasks, order, n, n_dev, next_slot, out, s);
    if (cols <= 128) return launch_qc<4>(variant, P, seq, tasks, order, n, n_dev, next_slot, out, s);
    if (cols <= 192) return launch_qc<6>(variant, P, seq, tasks, order, n, n_dev, next_slot, out, s);
    return launch_qc<8>(variant, P, seq, tasks, order, n, n_dev, next_slot, out, s);
}

}  // namespace bsw
