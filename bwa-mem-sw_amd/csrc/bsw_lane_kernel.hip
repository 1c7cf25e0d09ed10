/*
 * bsw_lane_kernel.hip — gfx950 kernel: ONE LANE PER EXTENSION (inter-task SIMD), for bins of
 * seeds with (nearly) equal query length — the batch manager's (qlen, tlen, w) bins.
 *
 * Each lane runs the scalar ksw_extend2 recurrence of one side of one seed verbatim
 * (sw_pe_array_sw_extend.v:1639-1705, CPU semantics of SURVEY.md §8a); the 64 lanes of a wave
 * walk DP row i together.  The lane's whole eh[] row lives in VGPRs, addressed statically
 * because the column loop is fully unrolled in 8-column blocks:
 *     B8  : two columns per VGPR, 8-bit h and e   (h0 + qlen*a <= 255 — the RTL's own int8
 *           datapath, sw_pe_array_sw_extend.v:101-108, but unsigned and range-checked by the host)
 *     !B8 : one column per VGPR, 16-bit h and e
 * Blocks no live lane touches are skipped with scalar branches, blocks inside every live lane's
 * [beg,end) run a mask-free "dense" body, the rest an exec-masked "edge" body.
 *   - score lookup: the query is held as three bit-planes per lane in LDS (code bit 0, bit 1, N); per row the target base
 *     selects a 32-column match mask, per cell v_bfe_i32 + v_bfi give +a / -b;
 *   - the packed target (16 bases per uint64) is staged per wave in LDS, 128 rows at a time;
 *   - only the first band try runs here.  A side that would need MAX_BAND_TRY's second pass
 *     (sw_pe_array_sw_extend.v:1837,1859) marks its seed for the wave-per-task kernel, which
 *     recomputes the seed from scratch (bsw_pair_finalize + redo list).
 * Eligibility (enforced by the host): bwa-style matrix (a on the diagonal, -b off it, one score for
 * every pair that involves an N), qlen < 8*QB, score range as above.
 */
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>
#include <utility>

#include "bsw_device.h"

namespace bsw {

template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ int ldpp(int old, int src)
{
    return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, BANK_MASK, false);
}
__device__ __forceinline__ int wave_max_all(int x)
{
    x = max(x, ldpp<0x111>(INT_MIN, x));
    x = max(x, ldpp<0x112>(INT_MIN, x));
    x = max(x, ldpp<0x114>(INT_MIN, x));
    x = max(x, ldpp<0x118>(INT_MIN, x));
    x = max(x, ldpp<0x142, 0xa>(INT_MIN, x));
    x = max(x, ldpp<0x143, 0xc>(INT_MIN, x));
    return __builtin_amdgcn_readlane(x, 63);
}

/* every 4th bit of a 64-bit word (bit `b` of each nibble) gathered into 16 contiguous bits */
__device__ __forceinline__ uint32_t nibble_plane(uint64_t w, int b)
{
    uint64_t x = (w >> b) & 0x1111111111111111ull;
    x = (x | (x >> 3)) & 0x0303030303030303ull;
    x = (x | (x >> 6)) & 0x000F000F000F000Full;
    x = (x | (x >> 12)) & 0x000000FF000000FFull;
    x = (x | (x >> 24)) & 0xFFFFull;
    return (uint32_t)x;
}

/* compile-time loop: every index is a constant from the front end on, so the eh[] register array is
 * scalarised by the first SROA pass instead of depending on the loop unroller */
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>)
{
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

struct lane_consts {
    int va, vnegb, vn;                               /* +a, -b, N score held in VGPRs (v_bfi takes one SGPR at most) */
    int oe_del, e_del, oe_ins, e_ins;
};

template <bool B8>
__device__ __forceinline__ uint32_t eh_put(uint32_t Pw, const int J, const uint32_t np16)
{
    if (!B8) return np16;
    return (J & 1) ? ((Pw & 0x0000ffffu) | (np16 << 16)) : ((Pw & 0xffff0000u) | np16);
}

/* One DP cell of column J for every lane whose [beg,end) contains J (EDGE) or for all live lanes (dense).
 * Pw is the VGPR holding eh[J]: B8 -> byte pair at bit (J&1)*16, else the whole word.
 * Returns true when the stored eh[J] is non-zero (for the next-row trimming, K8).
 * NQ: some lane of the wave has an N in this block of the query -> rnw marks those columns and they
 * score k.vn whatever the target base is (mat[.][4], sw_pe_array_sw_extend.v:1915-1940). */
template <int VAR, bool SYM, bool EDGE, bool B8, bool NQ>
__device__ __forceinline__ bool lane_cell(uint32_t &Pw, const int J, const uint32_t rmw, const uint32_t rnw,
                                          const int vmis, const lane_consts &k,
                                          const int beg, const int len, int &h1, int &f, int &mk)
{
    constexpr uint32_t HM = B8 ? 0xffu : 0xffffu;
    constexpr int HB = B8 ? 8 : 16;
    const int sh = B8 ? (J & 1) * 16 : 0;
    bool inr = true, nz = false;
    if (EDGE) inr = (unsigned)(J - beg) < (unsigned)len;
    if (inr) {
        const uint32_t p = Pw >> sh;
        const int hd = (int)(p & HM), e = (int)((p >> HB) & HM);     /* eh[j].h = H(i-1,j-1), eh[j].e */
        const int x = __builtin_amdgcn_sbfe(rmw, J & 31, 1);          /* -1 on match (q_j == t_i), else 0 */
        int s;                                                        /* +a | -b (:1915-1940): (x & a) | (~x & -b); hipcc expands */
        asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(s) : "v"(x), "v"(k.va), "v"(vmis));      /* the C form into 5 ops */
        if (NQ) {
            const int y = __builtin_amdgcn_sbfe(rnw, J & 31, 1);      /* -1 where q_j is N */
            asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(s) : "v"(y), "v"(k.vn), "v"(s));
        }
        int M = hd + s;                                               /* (:1797)                        */
        if (VAR == BSW_VARIANT_M) M = hd ? M : 0;
        const int h = max(max(M, e), f);                              /* (:1798,1809)                   */
        mk = max(mk, (h << 8) | (J & 63));                            /* row max of this 64-column group, ties -> later j */
        const int base = VAR == BSW_VARIANT_M ? M : h;
        const int tD = base - k.oe_del;
        const int tI = SYM ? tD : base - k.oe_ins;
        const int en = max(max(e - k.e_del, tD), 0);                  /* (:1866,1770-1771)              */
        f = max(max(f - k.e_ins, tI), 0);                             /* (:1863,1780-1781)              */
        const uint32_t np = ((uint32_t)en << HB) | (uint32_t)h1;      /* eh[j] = {e', H(i,j-1)} (:1776) */
        Pw = eh_put<B8>(Pw, J, np);
        h1 = h;
        nz = np != 0;
    }
    return nz;
}

#define BSW_LANE_TCHUNK 8       /* target words staged per wave in LDS = 128 DP rows */

template <int QB, int VAR, bool SYM, bool B8, int WPS>
__global__ __launch_bounds__(256, WPS) void bsw_lane_kernel(const bsw_dparams P, const int side,
                                                            const uint64_t *__restrict__ seq,
                                                            const bsw_dtask *__restrict__ tasks,
                                                            const uint32_t *__restrict__ order, const uint32_t n,
                                                            bsw_result *__restrict__ out)
{
    constexpr int QMAX = QB * 8;
    static_assert(QMAX <= BSW_LANE_QBINS, "a lane class holds at most 256 eh[] columns: the row-max key keeps the column in 8 bits and the binning sorts by query length in 256 bins");
    constexpr int NW = (QMAX + 31) / 32;
    constexpr int NP = B8 ? QMAX / 2 : QMAX;
    __shared__ uint64_t lds_t[4][BSW_LANE_TCHUNK][64];              /* [wave][word][lane]          */
    __shared__ uint32_t lds_q[4][3 * NW][64];                       /* query bit-planes: code bit 0, bit 1, N */
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t slot = blockIdx.x * 256u + threadIdx.x;
    const bool valid = slot < n;
    const uint32_t ti = valid ? order[slot] : order[0];
    int qlen, tlen, h0, w;
    uint32_t t_off, nblk = 0;
    {
        const bsw_dtask T = tasks[ti];
        int wlim;
        uint32_t q_off;
        if (side == 0) {
            qlen = T.lqlen; tlen = T.ltlen; wlim = T.wlim_l; q_off = T.lq_off; t_off = T.lt_off; h0 = T.h0;
        } else {
            qlen = T.rqlen; tlen = T.rtlen; wlim = T.wlim_r; q_off = T.rq_off; t_off = T.rt_off;
            h0 = T.lqlen > 0 ? out[ti].left.score : T.h0;             /* h0 = score after the left ext (:1671) */
        }
        if (!valid) tlen = 0;
        w = min(P.w, wlim);
        /* query -> three bit-planes (code bit 0, code bit 1, N), 32 columns per word, parked in LDS;
         * nblk: bit b set when some lane of the wave has an N in columns [8b, 8b+8) */
#pragma unroll
        for (int wd = 0; wd < NW; ++wd) {
            uint32_t q0 = 0, q1 = 0, q2 = 0;
#pragma unroll
            for (int hlf = 0; hlf < 2; ++hlf) {
                const int v = wd * 2 + hlf;
                if (v < (QMAX + 15) / 16) {
                    const uint64_t qw = (valid && v * 16 < qlen) ? seq[q_off + v] : 0ull;
                    q0 |= nibble_plane(qw, 0) << (hlf * 16);
                    q1 |= nibble_plane(qw, 1) << (hlf * 16);
                    q2 |= nibble_plane(qw, 2) << (hlf * 16);
                }
            }
            lds_q[wv][3 * wd][lane] = q0;
            lds_q[wv][3 * wd + 1][lane] = q1;
            lds_q[wv][3 * wd + 2][lane] = q2;
#pragma unroll
            for (int b = 0; b < 4; ++b)
                if (wd * 4 + b < QB && __builtin_amdgcn_ballot_w64(((q2 >> (8 * b)) & 0xffu) != 0) != 0) nblk |= 1u << (wd * 4 + b);
        }
    }
    lane_consts k;
    k.va = P.mat[0]; k.vnegb = P.mat[1]; k.vn = P.mat[24];
    asm volatile("" : "+v"(k.va), "+v"(k.vnegb), "+v"(k.vn));
    k.oe_del = P.o_del + P.e_del; k.e_del = P.e_del; k.oe_ins = P.o_ins + P.e_ins; k.e_ins = P.e_ins;
    const int o_del = P.o_del, e_del = P.e_del, zdrop = P.zdrop;
    const int ntw = (tlen + 15) >> 4;

    /* K2 first row, closed form: eh[0]=h0, eh[j]=max(h0-oe_ins-(j-1)e_ins,0), e=0 */
    uint32_t Pr[NP];
    static_for<NP>([&](auto ci) {
        constexpr int c = decltype(ci)::value;
        if (B8) {
            constexpr int j = 2 * c;
            const uint32_t lo = (uint32_t)(j == 0 ? h0 : max(h0 - k.oe_ins - (j - 1) * k.e_ins, 0));
            const uint32_t hi = (uint32_t)max(h0 - k.oe_ins - j * k.e_ins, 0);
            Pr[c] = lo | (hi << 16);
        } else {
            Pr[c] = (uint32_t)(c == 0 ? h0 : max(h0 - k.oe_ins - (c - 1) * k.e_ins, 0));
        }
    });

    int mx = h0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0, beg = 0, end = qlen;
    unsigned cells = 0;
    bool alive = tlen > 0;
    uint64_t tw = 0;

    for (int i = 0;; ++i) {
        const bool act = alive && i < tlen;
        if (__builtin_amdgcn_ballot_w64(act) == 0) break;
        if ((i & (BSW_LANE_TCHUNK * 16 - 1)) == 0) {                  /* stage the next 128 target bases of every lane */
            const int w0 = i >> 4;
#pragma unroll
            for (int x = 0; x < BSW_LANE_TCHUNK; ++x)
                lds_t[wv][x][lane] = (act && w0 + x < ntw) ? seq[t_off + w0 + x] : 0ull;
        }
        if ((i & 15) == 0) tw = lds_t[wv][(i >> 4) & (BSW_LANE_TCHUNK - 1)][lane];
        /* K3 band clamp, K4 column 0 (per lane) */
        const int nb = max(beg, i - w), ne = min(min(end, i + w + 1), qlen);
        beg = act ? nb : beg;
        end = act ? ne : end;
        const int len = max(end - beg, 0);
        /* live column range of the wave: blocks outside are skipped, blocks inside every lane's range run dense */
        const int jlo = -wave_max_all(act ? -beg : INT_MIN);
        const int jhi = wave_max_all(act ? end : INT_MIN);
        const int jbm = wave_max_all(act ? beg : INT_MIN);
        const int jem = -wave_max_all(act ? -end : INT_MIN);
        if (act) {
            const int tb = (int)((tw >> ((i & 15) * 4)) & 7);
            const uint32_t n0 = (uint32_t)((tb & 1) - 1), n1 = (uint32_t)(((tb >> 1) & 1) - 1);
            const uint32_t tn = (uint32_t)((tb >> 2) - 1);             /* 0 when the target base is N, else ~0 */
            const int vmis = tn ? k.vnegb : k.vn;                      /* a row against N scores mat[4][.] everywhere */
            uint32_t rm[NW];
#pragma unroll
            for (int wd = 0; wd < NW; ++wd)                            /* rm: 1 where q_j == t_i (and neither is N) */
                rm[wd] = (lds_q[wv][3 * wd][lane] ^ n0) & (lds_q[wv][3 * wd + 1][lane] ^ n1) & tn & ~lds_q[wv][3 * wd + 2][lane];
            int h1 = beg == 0 ? max(h0 - (o_del + e_del * (i + 1)), 0) : 0;
            /* Row max key and first/last non-zero column are kept per 64-column group, relative to the
             * group: VOP3 forms take no 32-bit literal on gfx9, so absolute column numbers > 64 would
             * each occupy a VGPR.  The groups are folded once per row. */
            constexpr int NG = (QMAX + 63) / 64;
            constexpr int NONE = 1 << 20;
            int f = 0, mkg[NG], fnzg[NG], lnzg[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g) { mkg[g] = -NONE; fnzg[g] = NONE; lnzg[g] = -NONE; }
            cells += (unsigned)len;
            static_for<QB>([&](auto blki) {
                constexpr int j0 = decltype(blki)::value * 8, g = j0 >> 6;
                if (j0 + 8 <= jlo || j0 > jhi) return;
                bool nz[8];
                const bool dense = j0 >= jbm && j0 + 8 <= jem;
                if ((nblk >> (j0 / 8)) & 1u) {                        /* rare: a query N somewhere in this block */
                    const uint32_t rnw = lds_q[wv][3 * (j0 >> 5) + 2][lane];   /* 1 where q_j is N; fetched on demand */
                    static_for<8>([&](auto ci) {
                        constexpr int J = j0 + decltype(ci)::value;
                        nz[J - j0] = lane_cell<VAR, SYM, true, B8, true>(Pr[B8 ? J / 2 : J], J, rm[J >> 5], rnw, vmis, k, beg, len, h1, f, mkg[g]);
                    });
                } else if (dense) {
                    static_for<8>([&](auto ci) {
                        constexpr int J = j0 + decltype(ci)::value;
                        nz[J - j0] = lane_cell<VAR, SYM, false, B8, false>(Pr[B8 ? J / 2 : J], J, rm[J >> 5], 0u, vmis, k, beg, len, h1, f, mkg[g]);
                    });
                } else {
                    static_for<8>([&](auto ci) {
                        constexpr int J = j0 + decltype(ci)::value;
                        nz[J - j0] = lane_cell<VAR, SYM, true, B8, false>(Pr[B8 ? J / 2 : J], J, rm[J >> 5], 0u, vmis, k, beg, len, h1, f, mkg[g]);
                    });
                }
                if (!dense) {
                    if (j0 + 8 > jem) {                               /* some lane's `end` is in this block: eh[end] = {0, h1} (:1775) */
                        static_for<8>([&](auto ci) {
                            constexpr int J = j0 + decltype(ci)::value;
                            uint32_t &Pw = Pr[B8 ? J / 2 : J];
                            Pw = J == end ? eh_put<B8>(Pw, J, (uint32_t)h1) : Pw;
                        });
                    }
                }
                /* first / last non-zero eh entry (K8) from the 8 compare masks of the block */
                int fb = NONE;
                static_for<8>([&](auto ci) {
                    constexpr int c = 7 - decltype(ci)::value;
                    fb = nz[c] ? ((j0 + c) & 63) : fb;
                });
                static_for<8>([&](auto ci) {
                    constexpr int c = decltype(ci)::value;
                    lnzg[g] = nz[c] ? ((j0 + c) & 63) : lnzg[g];
                });
                fnzg[g] = min(fnzg[g], fb);
            });
            int mk = mkg[0], fnz = fnzg[0], lnz = lnzg[0];
#pragma unroll
            for (int g = 1; g < NG; ++g) {
                mk = max(mk, mkg[g] + 64 * g);
                fnz = min(fnz, fnzg[g] + 64 * g);
                lnz = max(lnz, lnzg[g] + 64 * g);
            }
            /* K7 row tail */
            if (max(beg, end) == qlen) {                              /* ties -> later i (:1829-1833) */
                max_ie = gscore > h1 ? max_ie : i;
                gscore = max(gscore, h1);
            }
            const int m = mk < 0 ? 0 : (mk >> 8), mj = mk < 0 ? -1 : (mk & 255);
            bool stop = m == 0;                                       /* (:1942) */
            if (m > mx) {
                mx = m; max_i = i; max_j = mj;
                max_off = max(max_off, abs(mj - i));
            } else if (zdrop > 0) {
                const int di = i - max_i, dj = mj - max_j;
                const int pen = di > dj ? (di - dj) * e_del : (dj - di) * k.e_ins;
                stop = stop || (mx - m - pen > zdrop);
            }
            /* K8 next-row range (CPU semantics) */
            const int nbeg = fnz < end ? fnz : end;
            const int last = h1 != 0 ? end : (lnz >= 0 ? lnz : nbeg - 1);
            beg = nbeg;
            end = min(last + 2, qlen);
            alive = !stop;
        }
    }
    if (valid) {
        bsw_ext x;
        x.score = mx; x.qle = max_j + 1; x.tle = max_i + 1; x.gtle = max_ie + 1;
        x.gscore = gscore; x.max_off = max_off; x.aw = P.w; x.cells = cells;
        if (side == 0) out[ti].left = x; else out[ti].right = x;
    }
}

#if !defined(BSW_LANE_TU) || BSW_LANE_TU < 0
/* Pair-level decision for seeds whose sides came from the lane kernel (P2/P3:
 * sw_pe_array_proc_element.v:1593-1685).  A seed whose first band try does not satisfy the
 * MAX_BAND_TRY exit test goes to the redo list instead. */
__global__ __launch_bounds__(256) void bsw_pair_finalize(const bsw_dparams P, const bsw_dtask *__restrict__ tasks,
                                                         const uint32_t *__restrict__ order, const uint32_t n,
                                                         bsw_result *__restrict__ out,
                                                         uint32_t *__restrict__ redo, uint32_t *__restrict__ redo_cnt,
                                                         bsw_pair *__restrict__ pairs)
{
    const uint32_t slot = blockIdx.x * 256u + threadIdx.x;
    if (slot >= n) return;
    const uint32_t ti = order[slot];
    if (ti == BSW_ORDER_NONE) return;                       /* (a list's unused tail: bsw_binparams.nsplit) */
    const bsw_dtask T = tasks[ti];
    const bsw_result r = out[ti];
    bsw_pair_decide(P, T, ti, r.left, r.right, out, redo, redo_cnt, pairs);
}

/* BSW_RESULT_PAIR: the seeds the general kernels computed (their lists in `order`; n_dev != NULL: the redo list, counted on
 * the device) leave full records in `out`; their pair-level fields are copied into the dense array */
__global__ __launch_bounds__(256) void bsw_pairs_from_results(const uint32_t *__restrict__ order, const uint32_t n_host,
                                                              const uint32_t *__restrict__ n_dev,
                                                              const bsw_result *__restrict__ out, bsw_pair *__restrict__ pairs)
{
    const uint32_t n = n_dev ? *n_dev : n_host;
    for (uint32_t slot = blockIdx.x * 256u + threadIdx.x; slot < n; slot += gridDim.x * 256u) {
        const uint32_t ti = order[slot];
        const bsw_result *r = out + ti;
        bsw_pair pr;
        pr.tag = r->tag; pr.qb = r->qb; pr.qe = r->qe; pr.rb = r->rb; pr.re = r->re; pr.score = r->score; pr.truesc = r->truesc; pr.w = r->w;
        pairs[ti] = pr;
    }
}
#endif

/* lane classes: (bits per h/e value, 8-column blocks).  8-bit classes first.
 * The file is compiled once per class (-DBSW_LANE_TU=k, see Makefile) so the three sets of unrolled
 * kernels build in parallel; -DBSW_LANE_TU=-1 builds the class table, finalize kernel and dispatcher. */
template <int QB, bool B8, int WPS>
static hipError_t launch_lane_qb(int variant, bool sym, const bsw_dparams &P, int side, const uint64_t *seq,
                                 const bsw_dtask *tasks, const uint32_t *order, uint32_t n, bsw_result *out, hipStream_t s)
{
    const dim3 grid((n + 255u) / 256u), block(256);
    if (variant == BSW_VARIANT_M) {
        if (sym) hipLaunchKernelGGL((bsw_lane_kernel<QB, BSW_VARIANT_M, true, B8, WPS>), grid, block, 0, s, P, side, seq, tasks, order, n, out);
        else hipLaunchKernelGGL((bsw_lane_kernel<QB, BSW_VARIANT_M, false, B8, WPS>), grid, block, 0, s, P, side, seq, tasks, order, n, out);
    } else {
        if (sym) hipLaunchKernelGGL((bsw_lane_kernel<QB, BSW_VARIANT_H, true, B8, WPS>), grid, block, 0, s, P, side, seq, tasks, order, n, out);
        else hipLaunchKernelGGL((bsw_lane_kernel<QB, BSW_VARIANT_H, false, B8, WPS>), grid, block, 0, s, P, side, seq, tasks, order, n, out);
    }
    return hipGetLastError();
}

#define BSW_LANE_ARGS int variant, bool sym, const bsw_dparams &P, int side, const uint64_t *seq, const bsw_dtask *tasks, \
                      const uint32_t *order, uint32_t n, bsw_result *out, hipStream_t s
hipError_t launch_lane_c0(BSW_LANE_ARGS);
hipError_t launch_lane_c1(BSW_LANE_ARGS);
hipError_t launch_lane_c2(BSW_LANE_ARGS);

#ifndef BSW_LANE_TU
#define BSW_LANE_TU -2      /* single translation unit: everything */
#endif
#if BSW_LANE_TU == 0 || BSW_LANE_TU == -2
hipError_t launch_lane_c0(BSW_LANE_ARGS) { return launch_lane_qb<17, true, 4>(variant, sym, P, side, seq, tasks, order, n, out, s); }
#endif
#if BSW_LANE_TU == 1 || BSW_LANE_TU == -2
hipError_t launch_lane_c1(BSW_LANE_ARGS) { return launch_lane_qb<29, true, 3>(variant, sym, P, side, seq, tasks, order, n, out, s); }
#endif
#if BSW_LANE_TU == 2 || BSW_LANE_TU == -2
hipError_t launch_lane_c2(BSW_LANE_ARGS) { return launch_lane_qb<17, false, 2>(variant, sym, P, side, seq, tasks, order, n, out, s); }
#endif

#if BSW_LANE_TU < 0
/* lane classes, narrowest first (a side takes the first class of its value width that holds it).  `kind` names the kernel
 * that serves the class when the scoring parameters allow the two-seeds-per-lane formulation (lane2_params_ok); `fb` is the
 * one-seed-per-lane kernel (launch_lane_c0..c2) that serves it otherwise. */
enum { K_LANE2_9, K_LANE2_17, K_LANE2L_29, K_LANE16 };
struct lane_class_t { int bits, qb, kind, fb; };
/* The 72-column class (round 4): a 72-register row + the working set fit 168 VGPRs = THREE waves per SIMD (the 136-column
 * kernel holds two), issue ceiling 2.69 instead of 3.05 cycles per VALU instruction and SIMD (profiles/r3/ubench_fetch_align.txt)
 * and one more wave to hide the serial row head / tail.  Each reference PE serves any qlen <= 255 at one cell per clock
 * (sw_pe_array_sw_extend.v:101-102,144-148); here the short sides — half of the sides of a PE mixed-bin batch — get the
 * occupancy their narrow rows allow.  (Round 1 measured a 72-column class of the ONE-seed-per-lane kernel and dropped it:
 * that kernel gained no occupancy from it.)  BSW_NO_NARROW=1 removes the class (measurements). */
static const lane_class_t kLaneClassesAll[] = {{8, 9, K_LANE2_9, 0}, {8, 17, K_LANE2_17, 0}, {8, 29, K_LANE2L_29, 1}, {16, 17, K_LANE16, 2}};
static const lane_class_t *lane_classes(int *n)
{
    static const bool no_narrow = getenv("BSW_NO_NARROW") != nullptr;
    *n = no_narrow ? 3 : 4;
    return kLaneClassesAll + (no_narrow ? 1 : 0);
}

int lane_class_count() { int n; lane_classes(&n); return n; }
int lane_class_cols(int cls) { int n; return lane_classes(&n)[cls].qb * 8; }
int lane_class_bits(int cls) { int n; return lane_classes(&n)[cls].bits; }
bool lane_class_signals_tail(int cls) { int n; return cls >= 0 && lane_classes(&n)[cls].kind != K_LANE16; }

/* bsw_lane2_kernel.hip: two seeds per lane, packed 16-bit math, unrolled blocks — the 72-column class at three waves per
 * SIMD, the 136-column class at two; bsw_lane2l_kernel.hip: the same arithmetic with the blocks walked by a loop and the
 * row in AccVGPRs (232 columns, one wave per SIMD) */
bool lane2_params_ok(const bsw_dparams &P, int variant);
hipError_t launch_lane2(int qb, const bsw_dparams &P, int variant, int side, const uint64_t *seq, const bsw_dtask *tasks, const uint32_t *order,
                        uint32_t n, bsw_result *out, hipStream_t s, uint32_t *tail_flag, uint32_t *tail_target, const bsw_fin *fin);
hipError_t launch_lane2l(int qb, const bsw_dparams &P, int variant, int side, const uint64_t *seq, const bsw_dtask *tasks,
                         const uint32_t *order, uint32_t n, bsw_result *out, hipStream_t s, uint32_t *tail_flag, uint32_t *tail_target, const bsw_fin *fin);
/* bsw_lane2g_kernel.hip: the same arithmetic with a seed pair spread over a group of eight lanes — mid-sized chunks (bsw_fin.group) */
hipError_t launch_lane2g(int cols, const bsw_dparams &P, int variant, int side, const uint64_t *seq, const bsw_dtask *tasks, const uint32_t *order,
                         uint32_t n, bsw_result *out, hipStream_t s, const bsw_fin *fin);

static hipError_t launch_lane_k(int cls, int variant, const bsw_dparams &P, int side, const uint64_t *seq, const bsw_dtask *tasks,
                                const uint32_t *order, uint32_t n, bsw_result *out, hipStream_t s, uint32_t *&tail_flag, uint32_t *tail_target, const bsw_fin *fin);

/* which kernel serves the class under these parameters: 2 = bsw_lane2_kernel, 3 = bsw_lane2l_kernel, 1 = round 1's
 * one-seed-per-lane kernel (16-bit rows, N scores outside the packed range, gap penalties >= 256) */
static int lane_kernel_of(int cls, const bsw_dparams &P, int variant)
{
    int ncls;
    const lane_class_t &C = lane_classes(&ncls)[cls];
    if (C.kind != K_LANE16 && lane2_params_ok(P, variant)) {
        /* BSW_NO_LANE2L=1 selects round 1's one-seed-per-lane kernel for the 232-column class, BSW_LANE2L_NARROW=1 sends the
         * 136-column class through the looped kernel too (measurements; the unrolled kernel instantiated for 232 columns is
         * instruction-cache bound at one wave per SIMD: profiles/r3/lane2_wide_*) */
        static const bool nol = getenv("BSW_NO_LANE2L") != nullptr, narrow = getenv("BSW_LANE2L_NARROW") != nullptr;
        if (C.kind == K_LANE2L_29) return nol ? 1 : 3;
        return (C.kind == K_LANE2_17 && narrow) ? 3 : 2;
    }
    return 1;
}
/* the class's kernel finishes a seed in the epilogue of its last side (bsw_fin): the two-seeds-per-lane kernels do */
bool lane_class_finishes(int cls, const bsw_dparams &P, int variant)
{
    static const bool off = getenv("BSW_NO_FOLD") != nullptr;         /* (measurements: bsw_pair_finalize as a launch of its own) */
    return !off && lane_kernel_of(cls, P, variant) != 1;
}

hipError_t launch_lane(int cls, int variant, const bsw_dparams &P, int side, const uint64_t *seq, const bsw_dtask *tasks,
                       const uint32_t *order, uint32_t n, bsw_result *out, hipStream_t s, uint32_t *tail_flag, uint32_t *tail_target, const bsw_fin *fin)
{
    if (tail_target) *tail_target = 1u;
    hipError_t e = launch_lane_k(cls, variant, P, side, seq, tasks, order, n, out, s, tail_flag, tail_target, fin);
    /* nobody took the flag (an empty launch, a kernel that does not signal): it is raised behind the launch instead, so
     * that whoever waits for it never waits forever */
    if (e == hipSuccess && tail_flag) e = hipMemsetD32Async((hipDeviceptr_t)tail_flag, 1, 1, s);
    return e;
}

static hipError_t launch_lane_k(int cls, int variant, const bsw_dparams &P, int side, const uint64_t *seq, const bsw_dtask *tasks,
                                const uint32_t *order, uint32_t n, bsw_result *out, hipStream_t s, uint32_t *&tail_flag, uint32_t *tail_target, const bsw_fin *fin)
{
    if (n == 0) return hipSuccess;
    int ncls;
    const lane_class_t &C = lane_classes(&ncls)[cls];
    const int kern = lane_kernel_of(cls, P, variant);
    if (kern != 1 && fin && fin->group)                               /* (does not signal its tail: launch_lane raises the flag behind it) */
        return launch_lane2g(C.qb * 8, P, variant, side, seq, tasks, order, n, out, s, fin);
    if (kern != 1) {
        uint32_t *tf = tail_flag;
        tail_flag = nullptr;                                           /* (these kernels raise the flag themselves) */
        if (kern == 3) return launch_lane2l(C.kind == K_LANE2L_29 ? 29 : 17, P, variant, side, seq, tasks, order, n, out, s, tf, tail_target, fin);
        return launch_lane2(C.qb, P, variant, side, seq, tasks, order, n, out, s, tf, tail_target, fin);
    }
    const bool sym = P.o_del == P.o_ins && P.e_del == P.e_ins;
    switch (C.fb) {
    case 0: return launch_lane_c0(variant, sym, P, side, seq, tasks, order, n, out, s);
    case 1: return launch_lane_c1(variant, sym, P, side, seq, tasks, order, n, out, s);
    default: return launch_lane_c2(variant, sym, P, side, seq, tasks, order, n, out, s);
    }
}

hipError_t launch_finalize(const bsw_dparams &P, const bsw_dtask *tasks, const uint32_t *order, uint32_t n,
                           bsw_result *out, uint32_t *redo, uint32_t *redo_cnt, bsw_pair *pairs, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(bsw_pair_finalize, dim3((n + 255u) / 256u), dim3(256), 0, s, P, tasks, order, n, out, redo, redo_cnt, pairs);
    return hipGetLastError();
}

/* One wave that sleeps until *flag >= target OR A DEADLINE PASSES: what a stream runs in front of a launch that should not
 * start before the launch ahead of it in the chunk's chain has placed its last workgroup (enqueue_parts).
 * It polls every ~3 us and sleeps in between — the runtime's own stream wait (hipStreamWaitValue32 ->
 * __amd_rocclr_streamOpsWait) spins without a pause and takes the issue slots of the waves that share its SIMD: every
 * launch beside it got a straggler (a 1.5 ms half-batch launch took 2.4, gpurun_out/r7b).
 *
 * THE FLAG IS A SCHEDULING HINT, NOT A DEPENDENCY: every data dependency between the links of a chain is a stream event
 * (plan.dep -> ev_link in enqueue_parts).  A follower released early is still correct, it merely competes for wave slots
 * with its predecessor.  So the wait is BOUNDED: after BSW_WAIT_TICKS of the constant 100 MHz s_memrealtime counter
 * (20 ms; the longest launch a chain holds, a 232-column side of 1 M seeds, runs 7 ms) the wave gives up and the follower
 * starts.  That is what makes the chain safe wherever kernels are run one at a time (rocprofv3 --pmc, HIP_LAUNCH_BLOCKING,
 * AMD_SERIALIZE_KERNEL, a debugger): there the waiter is scheduled BEFORE the launch that would raise its word, finds the
 * word down, and returns after the deadline instead of sleeping for ever (the reference's TBB -> PE array -> RBB hand-off is
 * a hardware FSM that cannot wedge, tbb.v:110-123, rbb.v:219-224).  tests/test_gpu_lane.py runs the chain under both settings. */
#ifndef BSW_WAIT_TICKS
#define BSW_WAIT_TICKS 2000000ull
#endif
__global__ __launch_bounds__(64) void bsw_wait_count(const uint32_t *flag, const uint32_t target, uint32_t *expired)
{
    if (threadIdx.x == 0) {
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (__builtin_amdgcn_s_memrealtime() - t0 > BSW_WAIT_TICKS) {
                if (expired) __hip_atomic_fetch_add(expired, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            __builtin_amdgcn_s_sleep(127);
        }
    }
}

hipError_t launch_wait_count(const uint32_t *flag, uint32_t target, uint32_t *expired, hipStream_t s)
{
    hipLaunchKernelGGL(bsw_wait_count, dim3(1), dim3(64), 0, s, flag, target, expired);
    return hipGetLastError();
}

hipError_t launch_pairs_from_results(const uint32_t *order, uint32_t n, const uint32_t *n_dev, const bsw_result *out, bsw_pair *pairs, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    uint32_t blocks = (n + 255u) / 256u;
    if (n_dev && blocks > 256u) blocks = 256u;          /* the redo list is short: stride over it */
    hipLaunchKernelGGL(bsw_pairs_from_results, dim3(blocks), dim3(256), 0, s, order, n, n_dev, out, pairs);
    return hipGetLastError();
}
#endif

}  // namespace bsw
