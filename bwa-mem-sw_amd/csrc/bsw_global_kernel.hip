/*
 * bsw_global_kernel.hip — gfx950 kernel for SURVEY.md §8f row F4: bwa's banded GLOBAL alignment with CIGAR
 * (ksw_global2 / ksw_global in bwa's ksw.c; the next Smith-Waterman user after seed extension inside
 * mem_reg2aln -> bwa_gen_cigar2).  Not in the reference RTL — it lives in the host software named at
 * /root/reference/README.md:7-18 — so the CPU restatement oracle/ksw_global_ref.c is the parity anchor.
 *
 * One wavefront per alignment, row-synchronous like bsw_wave_kernel.hip: lane l owns eh[] columns l*C + c;
 * F(i,j) is an exclusive prefix max over the columns (DPP scan), H(i,j-1) arrives by a wave shift.  Unlike the
 * extension kernels this one is MEMORY-shaped: every DP cell leaves one direction byte (h | e<<2 | f<<4, bwa's
 * encoding) in the backtrack matrix z[tlen][min(qlen, 2w+1)] in HBM, written coalesced (a row's in-band columns
 * are contiguous bytes), and lane 0 walks it backwards to emit the CIGAR.  Algorithmic traffic = 1 B per cell +
 * the packed sequences; the roof is HBM write bandwidth.
 * Scores are int32 and follow bwa's arithmetic on MINUS_INF = -0x40000000 exactly (the direction bits of cells
 * fed by out-of-band values depend on it).
 */
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>

#include "bsw_device.h"
#include "bsw_stage.h"

namespace bsw {

namespace {

constexpr int GMINF = -0x40000000;
constexpr int GNEG = INT_MIN + (1 << 26);                  /* below anything the recurrence can produce, with headroom for -(j-1)*e_ins */

template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ int gdpp(int old, int src)
{
    return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, BANK_MASK, false);
}
__device__ __forceinline__ int gscan_max(int x)            /* inclusive max-scan over the 64 lanes */
{
    x = max(x, gdpp<0x111>(GNEG, x));
    x = max(x, gdpp<0x112>(GNEG, x));
    x = max(x, gdpp<0x114>(GNEG, x));
    x = max(x, gdpp<0x118>(GNEG, x));
    x = max(x, gdpp<0x142, 0xa>(GNEG, x));
    x = max(x, gdpp<0x143, 0xc>(GNEG, x));
    return x;
}

}  // namespace

template <int C>
__global__ __launch_bounds__(256) void bsw_global_kernel(const bsw_dparams P, const uint64_t *__restrict__ seq,
                                                         const bsw_gdtask *__restrict__ tasks, const uint32_t *__restrict__ order,
                                                         const uint32_t n, uint8_t *__restrict__ z,
                                                         uint32_t *__restrict__ cigars, const int max_cigar,
                                                         bsw_gresult *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const uint32_t slot = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (slot >= n) return;
    const uint32_t ti = order[slot];
    const bsw_gdtask T = tasks[ti];
    const int qlen = T.qlen, tlen = T.tlen, w = T.w;
    const int o_del = P.o_del, e_del = P.e_del, o_ins = P.o_ins, e_ins = P.e_ins;
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    const int jbase = lane * C;
    const int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;
    uint8_t *zt = z ? z + T.z_off : nullptr;

    /* query -> per-column score profile (any 5x5 matrix) */
    uint32_t prof_lo[C];
    int prof_hi[C];
    {
        const int nqw = (qlen + 15) >> 4;
        uint64_t qw = lane < nqw ? seq[T.q_off + lane] : 0ull;
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

    /* first row: eh[0] = {0,-inf}; eh[j].h = -(o_ins + e_ins*j) inside the band, -inf outside */
    int X[C], E[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int j = jbase + c;
        X[c] = j == 0 ? 0 : (j <= w ? -(o_ins + e_ins * j) : GMINF);
        E[c] = GMINF;
    }
    const int ntw = (tlen + 15) >> 4;
    uint32_t twl = 0, twh = 0, cur_lo = 0, cur_hi = 0;

    for (int i = 0; i < tlen; ++i) {
        if ((i & 1023) == 0) {                               /* coalesced refill: 64 words = 1024 target bases */
            const int wi = (i >> 4) + lane;
            const uint64_t tv = wi < ntw ? seq[T.t_off + wi] : 0ull;
            twl = (uint32_t)tv;
            twh = (uint32_t)(tv >> 32);
        }
        if ((i & 15) == 0) {
            const int src = (i >> 4) & 63;
            cur_lo = __builtin_amdgcn_readlane(twl, src);
            cur_hi = __builtin_amdgcn_readlane(twh, src);
        }
        int tb = (int)((((i & 8) ? cur_hi : cur_lo) >> ((i & 7) * 4)) & 7);
        tb = tb < 4 ? tb : 4;
        const int beg = i > w ? i - w : 0;
        const int end = i + w + 1 < qlen ? i + w + 1 : qlen;
        const int h1_init = beg == 0 ? -(o_del + e_del * (i + 1)) : GMINF;
        const int sh = (tb & 3) * 8;

        /* phase 1: M and the scan input G_k = M_k - oe_ins + k*e_ins of the in-band columns */
        int Mv[C], g[C];
        bool inr[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int j = jbase + c;
            inr[c] = j >= beg && j < end;
            const int s = tb < 4 ? (int)(int8_t)(prof_lo[c] >> sh) : prof_hi[c];
            Mv[c] = X[c] + s;
            g[c] = inr[c] ? Mv[c] - oe_ins + j * e_ins : GNEG;
        }
        /* phase 2: F(i,j) = max(MINUS_INF - (j-beg)e_ins, max_{beg<=k<j} G_k - (j-1)e_ins): exclusive prefix max */
        int pl[C];
        pl[0] = g[0];
#pragma unroll
        for (int c = 1; c < C; ++c) pl[c] = max(pl[c - 1], g[c]);
        const int incl = gscan_max(pl[C - 1]);
        const int carry = gdpp<0x138>(GNEG, incl);           /* wave_shr:1 -> exclusive */
        const int g_init = GMINF + (beg - 1) * e_ins;        /* f enters column beg as MINUS_INF */

        /* phase 3: H, direction bits, E' */
        int hv[C];
        uint32_t dbits[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int j = jbase + c;
            const int pex = max(c == 0 ? carry : max(carry, pl[c - 1]), g_init);
            const int f = pex - (j - 1) * e_ins;
            const int m = Mv[c], e = E[c];
            uint32_t d = m >= e ? 0u : 1u;
            int h = m >= e ? m : e;
            d = h >= f ? d : 2u;
            h = h >= f ? h : f;
            hv[c] = h;
            int t = m - oe_del;
            const int e2 = e - e_del;
            d |= e2 > t ? 1u << 2 : 0u;
            const int en = e2 > t ? e2 : t;
            t = m - oe_ins;
            d |= (f - e_ins) > t ? 2u << 4 : 0u;
            E[c] = inr[c] ? en : E[c];
            dbits[c] = d;
        }
        /* backtrack row: the in-band columns are contiguous bytes */
        if (zt) {
            uint8_t *zi = zt + (size_t)i * (size_t)n_col;
#pragma unroll
            for (int c = 0; c < C; ++c)
                if (inr[c]) zi[jbase + c - beg] = (uint8_t)dbits[c];
        }
        /* phase 4: eh[j].h <- H(i,j-1) for j in [beg,end]; eh[end].e <- -inf */
        const int hleft = gdpp<0x138>(0, hv[C - 1]);
#pragma unroll
        for (int c = C - 1; c >= 0; --c) {
            const int j = jbase + c;
            const bool wr = (j >= beg && j <= end) || j == end;      /* eh[end] is written even when the band has left the query (end < beg) */
            const int hp = c == 0 ? hleft : hv[c - 1];
            const int xn = (j == beg || end <= beg) ? h1_init : hp;
            X[c] = wr ? xn : X[c];
            E[c] = j == end ? GMINF : E[c];
        }
    }
    /* score = eh[qlen].h */
    const int lane_q = qlen / C, cq = qlen - lane_q * C;
    int xsel = X[0];
#pragma unroll
    for (int c = 1; c < C; ++c) xsel = cq == c ? X[c] : xsel;
    const int score = __builtin_amdgcn_readlane(xsel, lane_q);

    if (zt) __threadfence();                                 /* the wave's z stores are visible to lane 0's loads */
    if (lane == 0) {
        int n_cigar = 0;
        if (zt) {                                            /* backtrack from the last cell, ops pushed in reverse */
            uint32_t *cg = cigars + (size_t)ti * (size_t)max_cigar;
            uint32_t last = 0xffffffffu;
            int which = 0, i = tlen - 1, k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1;
            auto push = [&](uint32_t op, int len) {
                if (n_cigar == 0 || op != (last & 0xf)) {
                    if (n_cigar > 0 && n_cigar <= max_cigar) cg[n_cigar - 1] = last;
                    last = ((uint32_t)len << 4) | op;
                    ++n_cigar;
                } else last += (uint32_t)len << 4;
            };
            while (i >= 0 && k >= 0) {
                /* a band narrower than |qlen - tlen| cannot hold the path: bwa then walks through z entries of other
                 * cells (unspecified result); here such a step reads as "diagonal" and never leaves the matrix */
                const int col = k - (i > w ? i - w : 0);
                const uint8_t d = (col >= 0 && col < n_col) ? zt[(size_t)i * (size_t)n_col + (size_t)col] : (uint8_t)0;
                which = (d >> (which << 1)) & 3;
                if (which == 0) { push(0u, 1); --i; --k; }
                else if (which == 1) { push(2u, 1); --i; }
                else { push(1u, 1); --k; }
            }
            if (i >= 0) push(2u, i + 1);
            if (k >= 0) push(1u, k + 1);
            if (n_cigar > 0 && n_cigar <= max_cigar) cg[n_cigar - 1] = last;
            if (n_cigar <= max_cigar)
                for (int a = 0; a < n_cigar >> 1; ++a) { const uint32_t t = cg[a]; cg[a] = cg[n_cigar - 1 - a]; cg[n_cigar - 1 - a] = t; }
            else n_cigar = -n_cigar;                         /* did not fit: the caller retries with more room */
        }
        bsw_gresult r;
        r.score = score;
        r.n_cigar = n_cigar;
        out[ti] = r;
    }
}

static const int kGlobalClasses[] = {1, 2, 4, 8, 16};
int global_class_count() { return (int)(sizeof(kGlobalClasses) / sizeof(kGlobalClasses[0])); }
int global_class_cols(int cls) { return kGlobalClasses[cls] * 64; }

template <int C>
static hipError_t launch_gc(const bsw_dparams &P, const uint64_t *seq, const bsw_gdtask *tasks, const uint32_t *order, uint32_t n,
                            uint8_t *z, uint32_t *cigars, int max_cigar, bsw_gresult *out, hipStream_t s)
{
    hipLaunchKernelGGL((bsw_global_kernel<C>), dim3((n + 3u) / 4u), dim3(256), 0, s, P, seq, tasks, order, n, z, cigars, max_cigar, out);
    return hipGetLastError();
}

hipError_t launch_global(int cls, const bsw_dparams &P, const uint64_t *seq, const bsw_gdtask *tasks, const uint32_t *order, uint32_t n,
                         uint8_t *z, uint32_t *cigars, int max_cigar, bsw_gresult *out, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    switch (kGlobalClasses[cls]) {
    case 1: return launch_gc<1>(P, seq, tasks, order, n, z, cigars, max_cigar, out, s);
    case 2: return launch_gc<2>(P, seq, tasks, order, n, z, cigars, max_cigar, out, s);
    case 4: return launch_gc<4>(P, seq, tasks, order, n, z, cigars, max_cigar, out, s);
    case 8: return launch_gc<8>(P, seq, tasks, order, n, z, cigars, max_cigar, out, s);
    default: return launch_gc<16>(P, seq, tasks, order, n, z, cigars, max_cigar, out, s);
    }
}

}  // namespace bsw
