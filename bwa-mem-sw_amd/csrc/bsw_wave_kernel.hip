/*
 * bsw_wave_kernel.hip — gfx950 kernel: ONE WAVEFRONT PER SEED, row-synchronous banded
 * affine-gap extension (ksw_extend2) with the mem_chain2aln driver fused in.
 *
 * Replaces, per seed, what one RTL processing element does
 *   sw_pe_array_proc_element.v:1270-1446  (left ext, right ext, decision, 5-word record)
 *   sw_pe_array_sw_extend.v:1639-1705     (band-retry loop, row loop, II=1 cell pipeline)
 * with CPU (bwa) semantics — RTL quirks Q1-Q7 of SURVEY.md §8a are not reproduced.
 *
 * Mapping: lane l owns eh[] entries j = l*C + c, c < C (blocked layout),
 * resident in VGPRs for the whole extension (the RTL's 256x16b eh_arr BRAM,
 * sw_pe_array_sw_extend_eh_arr.v).  All 64 lanes process DP row i in the same step:
 *   - F(i,j) — the only intra-row dependency — is an exclusive prefix max of
 *     G_k = max(base_k - oe_ins, 0) + k*e_ins, done with a 6-step DPP scan;
 *   - eh[j].h <- H(i,j-1) is one wave_shr:1 DPP move;
 *   - row max / arg-max (ties -> later j) is a max over keys (h << 10 | j);
 *   - beg/end trimming, "j == qlen", m == 0, zdrop are wave-uniform (ballot / SALU);
 *   - writes are masked to [beg,end] so stale eh[] entries survive exactly as on the CPU.
 * The target is read coalesced (lane l holds packed word l of a 1024-base chunk) and
 * broadcast with v_readlane; the query becomes a per-column score profile in VGPRs.
 */
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>
#include <stdlib.h>

#include "bsw_device.h"

namespace bsw {

template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ int dpp_mov(int old, int src)
{
    return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, BANK_MASK, false);
}

/* inclusive max-scan over the 64 lanes (row_shr 1,2,4,8 + row_bcast 15/31) */
__device__ __forceinline__ int wave_scan_max(int x)
{
    x = max(x, dpp_mov<0x111>(INT_MIN, x));
    x = max(x, dpp_mov<0x112>(INT_MIN, x));
    x = max(x, dpp_mov<0x114>(INT_MIN, x));
    x = max(x, dpp_mov<0x118>(INT_MIN, x));
    x = max(x, dpp_mov<0x142, 0xa>(INT_MIN, x));
    x = max(x, dpp_mov<0x143, 0xc>(INT_MIN, x));
    return x;
}

constexpr int NEGV = -(1 << 29);                       /* "minus infinity" that survives -1023*e_ins */

__device__ __forceinline__ int sget(int v) { return __builtin_amdgcn_readfirstlane(v); }

struct side_out {
    int score, qle, tle, gtle, gscore, max_off, aw;
    unsigned cells;
};

template <int C, int VAR>
__device__ __forceinline__ void extend_side(const bsw_dparams &P, const uint64_t *__restrict__ seq,
                                            uint32_t q_off, uint32_t t_off, int qlen, int tlen, int wlim,
                                            int end_bonus_unused, int h0, int prev_score, int lane, side_out &so)
{
    (void)end_bonus_unused;
    const int o_del = P.o_del, e_del = P.e_del, o_ins = P.o_ins, e_ins = P.e_ins;
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    const int jbase = lane * C;

    /* ---- query -> per-column score profile (K6: sw_pe_array_sw_extend.v:1915-1940) ---- */
    uint32_t prof_lo[C];
    int prof_hi[C];
    {
        const int nqw = (qlen + 15) >> 4;
        uint64_t qw = lane < nqw ? seq[q_off + lane] : 0ull;
        uint32_t cp_lo[5];
        int cp_hi[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            cp_lo[q] = (uint32_t)(uint8_t)P.mat[q] | ((uint32_t)(uint8_t)P.mat[5 + q] << 8) |
                       ((uint32_t)(uint8_t)P.mat[10 + q] << 16) | ((uint32_t)(uint8_t)P.mat[15 + q] << 24);
            cp_hi[q] = P.mat[20 + q];
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int j = jbase + c;
            const uint64_t wv = __shfl(qw, (j >> 4) & 63);
            int qb = (int)((wv >> ((j & 15) * 4)) & 7);
            qb = (j < qlen && qb < 4) ? qb : 4;
            uint32_t lo = cp_lo[4];
            int hi = cp_hi[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                lo = qb == q ? cp_lo[q] : lo;
                hi = qb == q ? cp_hi[q] : hi;
            }
            prof_lo[c] = lo;
            prof_hi[c] = hi;
        }
    }

    const int ntw = (tlen + 15) >> 4;
    int score = prev_score, aw = P.w;
    int o_qle = 0, o_tle = 0, o_gtle = 0, o_gscore = -1, o_moff = 0;
    unsigned cells = 0;
    const int tries = P.max_band_try > 0 ? P.max_band_try : 1;

    for (int k = 0; k < tries; ++k) {                       /* P1 band retry (:1963) */
        const int prev = score;
        aw = P.w << k;
        const int w = min(aw, wlim);

        /* K2 first row, closed form (:1979,1957,1974,1818-1821) */
        int X[C], E[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int j = jbase + c;
            X[c] = j == 0 ? h0 : max(h0 - oe_ins - (j - 1) * e_ins, 0);
            E[c] = 0;
        }
        int mx = h0, max_i = -1, max_j = -1, max_ie = -1, gs = -1, moff = 0, beg = 0, end = qlen;
        uint32_t twl = 0, twh = 0;
        uint32_t cur_lo = 0, cur_hi = 0;

        for (int i = 0; i < tlen; ++i) {                    /* st4 row loop (:1891) */
            if ((i & 1023) == 0) {                          /* coalesced refill: 64 words = 1024 target bases */
                const int wi = (i >> 4) + lane;
                const uint64_t tv = wi < ntw ? seq[t_off + wi] : 0ull;
                twl = (uint32_t)tv;
                twh = (uint32_t)(tv >> 32);
            }
            if ((i & 15) == 0) {
                const int src = (i >> 4) & 63;
                cur_lo = __builtin_amdgcn_readlane(twl, src);
                cur_hi = __builtin_amdgcn_readlane(twh, src);
            }
            int ti = (int)((((i & 8) ? cur_hi : cur_lo) >> ((i & 7) * 4)) & 7);
            ti = ti < 4 ? ti : 4;

            /* K3 band clamp (:1803,1894-1897,1842,1898) */
            beg = max(beg, i - w);
            end = min(min(end, i + w + 1), qlen);
            /* K4 column 0, CPU semantics (:1795-1796,1835; Q4 avoided) */
            const int h1_init = beg == 0 ? max(h0 - (o_del + e_del * (i + 1)), 0) : 0;
            cells += (unsigned)max(end - beg, 0);

            /* ---- phase 1: per-column M, max(M,e), scan input ---- */
            int Mv[C], ht[C], g[C];
            bool inr[C];
            const int sh = (ti & 3) * 8;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const int j = jbase + c;
                inr[c] = (unsigned)(j - beg) < (unsigned)(end - beg) && end > beg;
                const int s = ti < 4 ? (int)(int8_t)(prof_lo[c] >> sh) : prof_hi[c];
                if (VAR == BSW_VARIANT_M) Mv[c] = X[c] ? X[c] + s : 0;
                else Mv[c] = X[c] + s;                      /* variant H (:1797) */
                ht[c] = max(Mv[c], E[c]);                   /* (:1798) */
                const int base = VAR == BSW_VARIANT_M ? Mv[c] : ht[c];
                g[c] = inr[c] ? max(base - oe_ins, 0) + j * e_ins : NEGV;
            }
            /* ---- phase 2: exclusive prefix max over columns (F recurrence, :1863,1780-1781) ---- */
            int pl[C];
            pl[0] = g[0];
#pragma unroll
            for (int c = 1; c < C; ++c) pl[c] = max(pl[c - 1], g[c]);
            const int incl = wave_scan_max(pl[C - 1]);
            const int carry = dpp_mov<0x138>(NEGV, incl);      /* wave_shr:1 -> exclusive */

            /* ---- phase 3: H, E', row max key ---- */
            int hv[C];
            int lkey = -1;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const int j = jbase + c;
                const int pex = c == 0 ? carry : max(carry, pl[c - 1]);
                const int f = max(pex - (j - 1) * e_ins, 0);
                hv[c] = max(ht[c], f);                      /* (:1809,1944) */
                const int base = VAR == BSW_VARIANT_M ? Mv[c] : hv[c];
                const int en = max(E[c] - e_del, max(base - oe_del, 0));   /* (:1866,1770-1771) */
                E[c] = inr[c] ? en : E[c];
                const int key = inr[c] ? ((hv[c] << BSW_KEY_BITS) | j) : -1;   /* ties -> later j (:1808,1816) */
                lkey = max(lkey, key);
            }
            const int mkey = __builtin_amdgcn_readlane(wave_scan_max(lkey), 63);
            const int mrow = mkey < 0 ? 0 : (mkey >> BSW_KEY_BITS);
            const int mj = mkey < 0 ? -1 : (mkey & ((1 << BSW_KEY_BITS) - 1));

            /* ---- phase 4: eh[j].h <- H(i,j-1) for j in [beg,end]; eh[end].e <- 0 (:1776,1775) ---- */
            const int hleft = dpp_mov<0x138>(0, hv[C - 1]);    /* lane l-1's last column */
            uint64_t nzb[C];
#pragma unroll
            for (int c = C - 1; c >= 0; --c) {
                const int j = jbase + c;
                const bool wr = j >= beg && j <= end;
                const int hp = c == 0 ? hleft : hv[c - 1];
                const int xn = j == beg ? h1_init : hp;
                X[c] = wr ? xn : X[c];
                E[c] = j == end ? 0 : E[c];
                nzb[c] = __builtin_amdgcn_ballot_w64(wr && ((X[c] | E[c]) != 0));
            }
            /* row tail scalars (K7) */
            const int lane_e = end / C, ce = end - lane_e * C;
            int xsel = X[0];
#pragma unroll
            for (int c = 1; c < C; ++c) xsel = ce == c ? X[c] : xsel;
            /* eh[end].h; when the range is empty past the band (beg > end) the CPU's h1 is h1_init */
            const int hlast = end < beg ? h1_init : __builtin_amdgcn_readlane(xsel, lane_e);
            const int jfin = max(beg, end);
            if (jfin == qlen) {                             /* (:1913,1941,1829-1833) ties -> later i */
                max_ie = gs > hlast ? max_ie : i;
                gs = max(gs, hlast);
            }
            if (mrow == 0) break;                           /* (:1942) */
            if (mrow > mx) {                                /* (:1959,1810,1845,1812-1813) */
                mx = mrow; max_i = i; max_j = mj;
                moff = max(moff, abs(mj - i));
            } else if (P.zdrop > 0) {                       /* C ABI only; RTL has no zdrop (Q3) */
                if (i - max_i > mj - max_j) {
                    if (mx - mrow - ((i - max_i) - (mj - max_j)) * e_del > P.zdrop) break;
                } else {
                    if (mx - mrow - ((mj - max_j) - (i - max_i)) * e_ins > P.zdrop) break;
                }
            }
            /* K8 next-row range, CPU semantics (Q5 avoided) */
            int first_nz = INT_MAX, last_nz = -1;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const uint64_t b = nzb[c];
                const int fl = b ? (int)__builtin_ctzll(b) * C + c : INT_MAX;
                const int ll = b ? (63 - (int)__builtin_clzll(b)) * C + c : -1;
                first_nz = min(first_nz, fl);
                last_nz = max(last_nz, ll);
            }
            const int nbeg = first_nz < end ? first_nz : end;
            const int last = last_nz >= 0 ? last_nz : nbeg - 1;
            beg = nbeg;
            end = min(last + 2, qlen);
        }
        score = mx;
        o_qle = max_j + 1; o_tle = max_i + 1; o_gtle = max_ie + 1; o_gscore = gs; o_moff = moff;   /* K9 */
        if (score == prev || moff < (aw >> 1) + (aw >> 2)) break;   /* (:1837,1859,1822) */
    }
    so.score = score; so.qle = o_qle; so.tle = o_tle; so.gtle = o_gtle; so.gscore = o_gscore;
    so.max_off = o_moff; so.aw = aw; so.cells = cells;
}


/* mem_chain2aln left/right driver for one seed (P2/P3: sw_pe_array_proc_element.v:1593-1685) */
template <int C, int VAR>
__global__ __launch_bounds__(256) void bsw_wave_kernel(const bsw_dparams P, const uint64_t *__restrict__ seq,
                                                       const bsw_dtask *__restrict__ tasks,
                                                       const uint32_t *__restrict__ order, const uint32_t n_host,
                                                       const uint32_t *__restrict__ n_dev,
                                                       bsw_result *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    /* n_dev != NULL: the seed count is produced on the device (redo list of the lane kernel);
     * the grid is then sized by an upper bound and strides over the list. */
    const uint32_t n = n_dev ? *n_dev : n_host;
    for (uint32_t slot = blockIdx.x * 4u + (uint32_t)sget((int)(threadIdx.x >> 6)); slot < n; slot += gridDim.x * 4u) {
        const uint32_t ti = order[slot];
        const bsw_dtask T = tasks[ti];

        side_out L, R;
        L.score = 0; L.qle = L.tle = L.gtle = 0; L.gscore = 0; L.max_off = 0; L.aw = P.w; L.cells = 0;
        R = L;
        int score = T.init_score, truesc, qb, rb, qe, re;
        if (T.lqlen > 0) {
            extend_side<C, VAR>(P, seq, T.lq_off, T.lt_off, T.lqlen, T.ltlen, T.wlim_l, 0, T.h0, score, lane, L);
            score = L.score;
            if (L.gscore <= 0 || L.gscore <= score - P.pen_clip5) {     /* local (:1672,1674-1675) */
                qb = T.qbeg - L.qle; rb = -L.tle; truesc = score;
            } else {                                                    /* to-end */
                qb = 0; rb = -L.gtle; truesc = L.gscore;
            }
        } else {
            score = truesc = T.h0; qb = 0; rb = 0;
        }
        const int sc0 = score;                                          /* h0 of the right side (:1671) */
        if (T.rqlen > 0) {
            extend_side<C, VAR>(P, seq, T.rq_off, T.rt_off, T.rqlen, T.rtlen, T.wlim_r, 0, sc0, score, lane, R);
            score = R.score;
            if (R.gscore <= 0 || R.gscore <= score - P.pen_clip3) {
                qe = R.qle; re = R.tle; truesc += score - sc0;
            } else {
                qe = T.rqlen; re = R.gtle; truesc += R.gscore - sc0;
            }
        } else {
            qe = 0; re = 0;
        }
        if (lane == 0) {
            bsw_result r;
            r.tag = T.tag; r.qb = qb; r.qe = qe; r.rb = rb; r.re = re;
            r.score = score; r.truesc = truesc; r.w = max(L.aw, R.aw);   /* P3 (:1684,1669) */
            r.left.score = L.score; r.left.qle = L.qle; r.left.tle = L.tle; r.left.gtle = L.gtle;
            r.left.gscore = L.gscore; r.left.max_off = L.max_off; r.left.aw = L.aw; r.left.cells = L.cells;
            r.right.score = R.score; r.right.qle = R.qle; r.right.tle = R.tle; r.right.gtle = R.gtle;
            r.right.gscore = R.gscore; r.right.max_off = R.max_off; r.right.aw = R.aw; r.right.cells = R.cells;
            out[ti] = r;
        }
    }
}

hipError_t launch_quad(int cols, int variant, const bsw_dparams &P, const uint64_t *seq, const bsw_dtask *tasks,
                       const uint32_t *order, uint32_t n, const uint32_t *n_dev, uint32_t *next_slot, bsw_result *out, hipStream_t s);

/* columns per lane of the register kernels (64 .. 1 024 columns), then the two LDS-row classes of bsw_long_kernel.hip (2 048 and
 * 8 192 columns: queries up to BSW_MAX_QLEN = 8 191) */
static const int kWaveClasses[] = {1, 2, 3, 4, 8, 16, 32, 128};
hipError_t launch_long(int cols, int variant, const bsw_dparams &P, const uint64_t *seq, const bsw_dtask *tasks,
                       const uint32_t *order, uint32_t n, const uint32_t *n_dev, bsw_result *out, hipStream_t s);

int wave_class_count() { return (int)(sizeof(kWaveClasses) / sizeof(kWaveClasses[0])); }
int wave_class_cols(int cls) { return kWaveClasses[cls] * 64; }

template <int C>
static hipError_t launch_c(int variant, const bsw_dparams &P, const uint64_t *seq, const bsw_dtask *tasks,
                           const uint32_t *order, uint32_t n, const uint32_t *n_dev, bsw_result *out, hipStream_t s)
{
    uint32_t blocks = (n + 3u) / 4u;
    if (n_dev && blocks > 8192u) blocks = 8192u;        /* device-side count: stride over the list */
    const dim3 grid(blocks), block(256);
    if (variant == BSW_VARIANT_M)
        hipLaunchKernelGGL((bsw_wave_kernel<C, BSW_VARIANT_M>), grid, block, 0, s, P, seq, tasks, order, n, n_dev, out);
    else
        hipLaunchKernelGGL((bsw_wave_kernel<C, BSW_VARIANT_H>), grid, block, 0, s, P, seq, tasks, order, n, n_dev, out);
    return hipGetLastError();
}

/* n = seed count (or an upper bound of *n_dev when n_dev != NULL); next_slot = a zeroed device word for the four-seeds-per-
 * wavefront kernel's work counter (NULL: the one-wavefront-per-seed kernel runs the class) */
hipError_t launch_wave(int cls, int variant, const bsw_dparams &P, const uint64_t *seq, const bsw_dtask *tasks,
                       const uint32_t *order, uint32_t n, const uint32_t *n_dev, uint32_t *next_slot, bsw_result *out, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    /* Four seeds per wavefront, one 16-lane DPP row each (bsw_quad_kernel.hip), where it is the faster of the two general
     * kernels: classes of 192 / 256 columns (queries of 128 - 255 bases) from 8 192 seeds up.  Measured, device-resident
     * (gpurun_out/r4r, r4s2): 131 x 257 seeds 1.23x at 8 k, 1.43x at 16 k, 1.53x at 128 k seeds — but a lone four-seed wave
     * walks a row in 1.15 us where a one-seed wave needs 0.5, so below ~6 k seeds (the GPU not full) this file's kernel
     * wins; and on PE mixed bins (sides of 1 - 131 bases, most of them one 64-column wave class wide) the four-seed rows'
     * fixed cost per row leaves no gain at any size (0.41x at 1 k ... 0.96x at 128 k seeds).  BSW_QUAD=1 sends every class
     * up to 256 columns through it whatever the size (tests, measurements), BSW_QUAD=0 none. */
    static const int quad_mode = getenv("BSW_QUAD") ? atoi(getenv("BSW_QUAD")) : -1;
    const int cols = kWaveClasses[cls] * 64;
    if (cols > 1024) return launch_long(cols, variant, P, seq, tasks, order, n, n_dev, out, s);
    /* (round 5 measured a third general kernel — one wavefront per seed whose lanes FOLLOW the live band, the eh[] row in LDS:
     * tools/experiments/bsw_band_kernel.hip, bit-exact, 95 VALU lane-instructions per cell against 129 here — at 6.49 ms per
     * 131 072 seeds of 131 x 257 against 7.00 here and 4.57 for the four-seed kernel, slower on PE mixed seeds and on a lone
     * scalar call: profiles/r5/general_kernels_band_experiment.txt.  Not built into the library.) */
    const bool quad = next_slot && cols <= 256 && (quad_mode == 1 || (quad_mode < 0 && cols >= 192 && !n_dev && n >= 8192u));
    if (quad) return launch_quad(cols, variant, P, seq, tasks, order, n, n_dev, next_slot, out, s);
    switch (kWaveClasses[cls]) {
    case 1: return launch_c<1>(variant, P, seq, tasks, order, n, n_dev, out, s);
    case 2: return launch_c<2>(variant, P, seq, tasks, order, n, n_dev, out, s);
    case 3: return launch_c<3>(variant, P, seq, tasks, order, n, n_dev, out, s);
    case 4: return launch_c<4>(variant, P, seq, tasks, order, n, n_dev, out, s);
    case 8: return launch_c<8>(variant, P, seq, tasks, order, n, n_dev, out, s);
    default: return launch_c<16>(variant, P, seq, tasks, order, n, n_dev, out, s);
    }
}

}  // namespace bsw
