/*
 * bsw_long_kernel.hip — gfx950 kernel: ONE WAVEFRONT PER SEED, LANES FOLLOW THE LIVE BAND, eh[] row in LDS.  The general path for
 * queries of 1 024 to 8 191 bases (any 5x5 matrix, int32 scores, band retries in-kernel): what lifts BSW_MAX_QLEN so that the
 * drop-in ksw_extend2 never fails on a long read (bwa's routine has no such limit; the RTL's is 255,
 * sw_pe_array_sw_extend.v:101-102).
 *
 * Replaces, per seed, what one RTL processing element does
 *   sw_pe_array_proc_element.v:1270-1446  (left ext, right ext, decision, 5-word record)
 *   sw_pe_array_sw_extend.v:1639-1705     (band-retry loop, row loop, II=1 cell pipeline)
 * with CPU (bwa) semantics — RTL quirks Q1-Q7 of SURVEY.md §8a are not reproduced.
 *
 * bsw_wave_kernel.hip keeps the eh[] row in registers, lane l on the FIXED columns l*C .. l*C + C - 1: sixteen columns per lane
 * at 1 024 columns is where the registers end.  Here lane l of chunk k works on column  beg + 64 k + l  of the CURRENT row: the
 * lanes follow [beg, end] as it slides along the diagonal, a row runs ceil((end - beg + 1) / 64) chunks, and the row lives in
 * LDS, one 16-byte record per column — {h, e, the column's score profile for target bases ACGT packed 4 x int8, its score
 * against a target N} — read with one ds_read_b128 and written back with one ds_write_b64 per chunk (the RTL's PE, too,
 * touches only eh[beg .. end] of its eh_arr BRAM per row: sw_pe_array_sw_extend.v:1772-1799).  Entries outside [beg, end]
 * keep whatever they held, exactly as on the CPU.  LDS operations of one wave complete in order: no barrier.
 *   - F(i,j) — the only intra-row dependency — is an exclusive prefix max of G_j = max(base_j - oe_ins, 0) + j*e_ins
 *     (valid because o_ins >= 0): a 6-step DPP scan per chunk, the carry between chunks a scalar;
 *   - eh[j].h <- H(i,j-1) is one wave_shr:1 DPP move (lane 0 takes the previous chunk's last column);
 *   - the row maximum and its LAST column (ties -> later j) are two wave maxima (a packed score | column key would not fit
 *     32 bits at 8 191 columns and scores up to 2^20);
 *   - beg / end, "j == qlen", m == 0, zdrop, first / last non-zero entry are wave-uniform (ballots, SALU).
 * Round 5 built this mapping as an experiment for the 64 - 1 023 base classes (tools/experiments/bsw_band_kernel.hip: bit-exact,
 * slower there than the register kernels); round 6 uses it where no register kernel can go.
 */
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>
#include <stdlib.h>

#include "bsw_device.h"

namespace bsw {

namespace {

template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ int bdpp(int old, int src)
{
    return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, BANK_MASK, false);
}

/* inclusive max-scan over the 64 lanes (row_shr 1,2,4,8 + row_bcast 15/31) */
__device__ __forceinline__ int band_scan_max(int x)
{
    x = max(x, bdpp<0x111>(INT_MIN, x));
    x = max(x, bdpp<0x112>(INT_MIN, x));
    x = max(x, bdpp<0x114>(INT_MIN, x));
    x = max(x, bdpp<0x118>(INT_MIN, x));
    x = max(x, bdpp<0x142, 0xa>(INT_MIN, x));
    x = max(x, bdpp<0x143, 0xc>(INT_MIN, x));
    return x;
}

constexpr int NEGB = -(1 << 29);                       /* "minus infinity": 8 191 columns x e_ins <= 4 096 = 2^25 below it is still far from INT_MIN */

struct band_out {
    int score, qle, tle, gtle, gscore, max_off, aw;
    unsigned cells;
};

/* one side of one seed; row = this wave's LDS row, at least qlen + 1 + 64 records (the last chunk of a row may reach 63
 * columns past `end`; those records are read, never used, and written back unchanged) */
template <int VAR>
__device__ __forceinline__ void band_side(const bsw_dparams &P, const uint64_t *__restrict__ seq, uint32_t q_off, uint32_t t_off,
                                          int qlen, int tlen, int wlim, int h0, int prev_score, int lane, uint4 *row, band_out &so)
{
    const int o_del = P.o_del, e_del = P.e_del, o_ins = P.o_ins, e_ins = P.e_ins;
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    const int ncol = ((qlen + 1 + 63) & ~63) + 64;      /* records this side touches */

    /* ---- query -> per-column score profile (K6: sw_pe_array_sw_extend.v:1915-1940), kept in the row's records ---- */
    {
        uint32_t cp_lo[5];
        int cp_hi[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            cp_lo[q] = (uint32_t)(uint8_t)P.mat[q] | ((uint32_t)(uint8_t)P.mat[5 + q] << 8) |
                       ((uint32_t)(uint8_t)P.mat[10 + q] << 16) | ((uint32_t)(uint8_t)P.mat[15 + q] << 24);
            cp_hi[q] = P.mat[20 + q];
        }
        for (int j = lane; j < ncol; j += 64) {
            int qb = 4;
            if (j < qlen) qb = (int)((seq[q_off + (uint32_t)(j >> 4)] >> ((j & 15) * 4)) & 7);
            qb = qb < 4 ? qb : 4;
            uint32_t lo = cp_lo[4];
            int hi = cp_hi[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                lo = qb == q ? cp_lo[q] : lo;
                hi = qb == q ? cp_hi[q] : hi;
            }
            row[j] = make_uint4(0u, 0u, lo, (uint32_t)hi);
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

        /* K2 first row, closed form (:1979,1957,1974,1818-1821): h and e of every record, the profile stays */
        for (int j = lane; j < ncol; j += 64) {
            const int x = j == 0 ? h0 : (j <= qlen ? max(h0 - oe_ins - (j - 1) * e_ins, 0) : 0);
            *(uint2 *)&row[j] = make_uint2((uint32_t)x, 0u);
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
            const int sh = (ti & 3) * 8;

            /* K3 band clamp (:1803,1894-1897,1842,1898) */
            beg = max(beg, i - w);
            end = min(min(end, i + w + 1), qlen);
            /* K4 column 0, CPU semantics (:1795-1796,1835; Q4 avoided) */
            const int h1_init = beg == 0 ? max(h0 - (o_del + e_del * (i + 1)), 0) : 0;
            cells += (unsigned)max(end - beg, 0);

            /* ---- the row's chunks: columns beg + 64 c + lane, c = 0 .. (end - beg) / 64 (the entry eh[end] included) ---- */
            int lh = -1, lj = -1;                           /* row max of this lane's columns and its last column */
            int carry = NEGB;                               /* max G of every column left of the chunk */
            int hprev = 0;                                  /* H(i, j - 1) entering the chunk (chunk 0: column beg takes h1_init) */
            int first_nz = INT_MAX, last_nz = -1;
            int hlast = h1_init;                            /* eh[end].h; when the range is empty past the band (beg > end) the CPU's h1 is h1_init */
            const int nchunk = end >= beg ? ((end - beg) >> 6) + 1 : 0;
            for (int c = 0; c < nchunk; ++c) {
                const int j0 = beg + 64 * c, j = j0 + lane;
                const uint4 rec = row[j];
                int X = (int)rec.x, E = (int)rec.y;
                const bool inr = j < end, wr = j <= end;
                const int s = ti < 4 ? (int)(int8_t)(rec.z >> sh) : (int)rec.w;
                int Mv;
                if (VAR == BSW_VARIANT_M) Mv = X ? X + s : 0;
                else Mv = X + s;                            /* variant H (:1797) */
                const int ht = max(Mv, E);                  /* (:1798) */
                const int gbase = VAR == BSW_VARIANT_M ? Mv : ht;
                const int g = inr ? max(gbase - oe_ins, 0) + j * e_ins : NEGB;
                /* exclusive prefix max over the columns left of j (F recurrence, :1863,1780-1781) */
                const int incl = band_scan_max(g);
                const int pex = max(bdpp<0x138>(NEGB, incl), carry);        /* wave_shr:1 -> exclusive; lane 0: the chunks before */
                const int f = max(pex - (j - 1) * e_ins, 0);
                const int hv = max(ht, f);                  /* (:1809,1944) */
                const int ebase = VAR == BSW_VARIANT_M ? Mv : hv;
                const int en = max(E - e_del, max(ebase - oe_del, 0));      /* (:1866,1770-1771) */
                E = inr ? en : E;
                if (inr && hv >= lh) { lh = hv; lj = j; }                  /* ties -> later j (:1808,1816): a lane's columns ascend */
                /* eh[j].h <- H(i,j-1) for j in [beg,end]; eh[end].e <- 0 (:1776,1775) */
                const int hp = bdpp<0x138>(hprev, hv);      /* lane l-1's column; lane 0: the previous chunk's last one */
                const int xn = j == beg ? h1_init : hp;
                X = wr ? xn : X;
                E = j == end ? 0 : E;
                *(uint2 *)&row[j] = make_uint2((uint32_t)X, (uint32_t)E);
                const uint64_t nzb = __builtin_amdgcn_ballot_w64(wr && ((X | E) != 0));
                if (nzb) {
                    if (first_nz == INT_MAX) first_nz = j0 + (int)__builtin_ctzll(nzb);
                    last_nz = j0 + 63 - (int)__builtin_clzll(nzb);
                }
                carry = max(carry, __builtin_amdgcn_readlane(incl, 63));
                hprev = __builtin_amdgcn_readlane(hv, 63);
                if ((unsigned)(end - j0) < 64u) hlast = __builtin_amdgcn_readlane(X, end - j0);
            }
            const int mtop = __builtin_amdgcn_readlane(band_scan_max(lh), 63);
            const int mjt = __builtin_amdgcn_readlane(band_scan_max(lh == mtop ? lj : -1), 63);
            const int mrow = mtop < 0 ? 0 : mtop;
            const int mj = mtop < 0 ? -1 : mjt;

            /* row tail scalars (K7) */
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

}  // namespace

/* mem_chain2aln left/right driver for one seed (P2/P3: sw_pe_array_proc_element.v:1593-1685); cols = LDS records per wave */
template <int VAR, int WPB>
__global__ __launch_bounds__(64 * WPB) void bsw_long_kernel(const bsw_dparams P, const uint64_t *__restrict__ seq,
                                                       const bsw_dtask *__restrict__ tasks,
                                                       const uint32_t *__restrict__ order, const uint32_t n_host,
                                                       const uint32_t *__restrict__ n_dev, const int cols,
                                                       bsw_result *__restrict__ out)
{
    extern __shared__ uint4 long_rows[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint4 *row = long_rows + (size_t)wv * (size_t)cols;
    /* n_dev != NULL: the seed count is produced on the device (redo list of the lane kernels);
     * the grid is then sized by an upper bound and strides over the list. */
    const uint32_t n = n_dev ? *n_dev : n_host;
    for (uint32_t slot = blockIdx.x * (uint32_t)WPB + (uint32_t)wv; slot < n; slot += gridDim.x * (uint32_t)WPB) {
        const uint32_t ti = order[slot];
        const bsw_dtask T = tasks[ti];

        band_out L, R;
        L.score = 0; L.qle = L.tle = L.gtle = 0; L.gscore = 0; L.max_off = 0; L.aw = P.w; L.cells = 0;
        R = L;
        int score = T.init_score, truesc, qb, rb, qe, re;
        if (T.lqlen > 0) {
            band_side<VAR>(P, seq, T.lq_off, T.lt_off, T.lqlen, T.ltlen, T.wlim_l, T.h0, score, lane, row, L);
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
            band_side<VAR>(P, seq, T.rq_off, T.rt_off, T.rqlen, T.rtlen, T.wlim_r, sc0, score, lane, row, R);
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

/* cols = eh[] columns of the seed class (1 024 < cols <= 8 192); n = seed count (or an upper bound of *n_dev).  Up to 2 048
 * columns four wavefronts share a workgroup's LDS (4 x 2 176 records x 16 B = 136 KB), beyond that a wavefront has it alone. */
template <int VAR, int WPB>
static hipError_t launch_long_t(int rec, const bsw_dparams &P, const uint64_t *seq, const bsw_dtask *tasks, const uint32_t *order,
                                uint32_t n, const uint32_t *n_dev, bsw_result *out, hipStream_t s)
{
    const size_t lds = (size_t)rec * 16u * (size_t)WPB;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&bsw_long_kernel<VAR, WPB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr != hipSuccess) return attr;
    uint32_t blocks = (n + (uint32_t)WPB - 1u) / (uint32_t)WPB;
    if (n_dev && blocks > 8192u) blocks = 8192u;                        /* device-side count: stride over the list */
    hipLaunchKernelGGL((bsw_long_kernel<VAR, WPB>), dim3(blocks), dim3(64 * WPB), lds, s, P, seq, tasks, order, n, n_dev, rec, out);
    return hipGetLastError();
}

hipError_t launch_long(int cols, int variant, const bsw_dparams &P, const uint64_t *seq, const bsw_dtask *tasks,
                       const uint32_t *order, uint32_t n, const uint32_t *n_dev, bsw_result *out, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    const int rec = ((cols + 63) & ~63) + 64;                          /* LDS records per wave */
    const bool m = variant == BSW_VARIANT_M;
    if (cols <= 2048) return m ? launch_long_t<BSW_VARIANT_M, 4>(rec, P, seq, tasks, order, n, n_dev, out, s) : launch_long_t<BSW_VARIANT_H, 4>(rec, P, seq, tasks, order, n, n_dev, out, s);
    return m ? launch_long_t<BSW_VARIANT_M, 1>(rec, P, seq, tasks, order, n, n_dev, out, s) : launch_long_t<BSW_VARIANT_H, 1>(rec, P, seq, tasks, order, n, n_dev, out, s);
}

}  // namespace bsw
