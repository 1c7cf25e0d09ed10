/*
 * ksw_align_ref.c — CPU ORACLE (test infrastructure, NOT product code) for the second half of SURVEY.md §8f row F4:
 * bwa's striped local alignment ksw_align / ksw_align2 (ksw_u8, ksw_i16), the Smith-Waterman bwa-mem's mate rescue
 * (mem_matesw) calls.
 *
 * PARITY UNPINNED: ksw_align lives in the host software of the reference (peterpengwei/bwa-mem-quickassist,
 * bwa-0.7.8/ksw.c, named at /root/reference/README.md:7-18), which is not in this image; the RTL does not implement it.
 * This is a restatement of the published algorithm (Farrar's striped SW as bwa's ksw.c words it), and a LITERAL one:
 * the 16 (8-bit) or 8 (16-bit) SSE2 lanes are emulated one by one, because the outputs depend on the striping —
 *   - query position k sits in vector k % slen, lane k / slen (slen = ceil(qlen / lanes));
 *   - E(i+1,j) is taken from H(i,j) BEFORE the lazy-F correction ("we disallow adjacent insertion and then deletion");
 *   - the lazy-F loop runs at most 16 rounds and stops as soon as no lane's F exceeds H - oe_ins;
 *   - 8-bit scores saturate, the run stops at gmax + shift >= 255 and reports score 255;
 *   - qe is the first maximum of the kept H column in MEMORY order (vector-major), not in query order;
 *   - the sub-optimal list b[] merges a row into the previous entry only if that entry's stored row is i - 1.
 * Pinned by analytic KATs, score bounds against an independent numpy local DP (oracle/py/full_dp.py) and the
 * re-scoring of the reported end points in tests/test_oracle_align.py.
 */
#include "ksw_extend_ref.h"

#include <stdlib.h>
#include <string.h>

#define A_XBYTE  0x10000
#define A_XSTOP  0x20000
#define A_XSUBO  0x40000
#define A_XSTART 0x80000

typedef struct { int score, te, qe, score2, te2, tb, qb; } kswr_ref_t;    /* = bwa's kswr_t */

static const kswr_ref_t g_defr = {0, -1, -1, -1, -1, -1, -1};

static inline int sat_add_u8(int a, int b) { int s = a + b; return s > 255 ? 255 : s; }
static inline int sat_sub_u(int a, int b) { return a > b ? a - b : 0; }                /* _mm_subs_epu8 / _mm_subs_epu16 */
static inline int sat_add_i16(int a, int b) { int s = a + b; return s > 32767 ? 32767 : (s < -32768 ? -32768 : s); }
static inline int imax2(int a, int b) { return a > b ? a : b; }

/* One run of ksw_u8 (size 1, 16 lanes) or ksw_i16 (size 2, 8 lanes).  cells (optional) += qlen * rows evaluated. */
static kswr_ref_t ksw_vec_ref(int size, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat,
                              int o_del, int e_del, int o_ins, int e_ins, int xtra, uint64_t *cells)
{
    const int P = size == 1 ? 16 : 8;
    const int slen = (qlen + P - 1) / P;
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    int a, i, j, k, l, shift, mx, te = -1, gmax = 0, minsc, endsc, n_b = 0, m_b = 0;
    uint64_t *b = 0;
    kswr_ref_t r = g_defr;
    /* ksw_qinit: shift = -(smallest score), max = largest score (never below 0) */
    int8_t smin = 127, smax = 0;
    for (a = 0; a < m * m; ++a) { if (mat[a] < smin) smin = mat[a]; if (mat[a] > smax) smax = mat[a]; }
    mx = smax;
    shift = (256 - (int)(uint8_t)smin) & 0xff;
    const size_t nv = (size_t)(slen > 0 ? slen : 1) * (size_t)P;
    int *qp = (int *)calloc(nv * (size_t)m, sizeof(int));                       /* [a][j][lane] */
    int *H0 = (int *)calloc(nv, sizeof(int)), *H1 = (int *)calloc(nv, sizeof(int));
    int *E = (int *)calloc(nv, sizeof(int)), *Hmax = (int *)calloc(nv, sizeof(int));
    for (a = 0; a < m; ++a)
        for (j = 0; j < slen; ++j)
            for (l = 0; l < P; ++l) {
                k = j + l * slen;
                const int s = k >= qlen ? 0 : mat[a * m + (query[k] < m ? query[k] : m - 1)];
                qp[((size_t)a * slen + j) * P + l] = size == 1 ? ((s + shift) & 0xff) : s;
            }
    minsc = (xtra & A_XSUBO) ? xtra & 0xffff : 0x10000;
    endsc = (xtra & A_XSTOP) ? xtra & 0xffff : 0x10000;
    for (i = 0; i < tlen; ++i) {
        int f[16], h[16], mxv[16], imax = 0, stop = 0;
        const int tb = target[i] < m ? target[i] : m - 1;
        const int *S = qp + (size_t)tb * slen * P;
        if (cells) *cells += (uint64_t)qlen;
        for (l = 0; l < P; ++l) { f[l] = 0; mxv[l] = 0; }
        h[0] = 0;                                                              /* h = H(i-1, -1): the last vector shifted by one lane */
        for (l = 1; l < P; ++l) h[l] = slen ? H0[(size_t)(slen - 1) * P + l - 1] : 0;
        for (j = 0; j < slen; ++j) {
            for (l = 0; l < P; ++l) {
                int hh, e, t;
                if (size == 1) hh = sat_sub_u(sat_add_u8(h[l], S[(size_t)j * P + l]), shift);
                else hh = sat_add_i16(h[l], S[(size_t)j * P + l]);
                e = E[(size_t)j * P + l];
                hh = imax2(imax2(hh, e), f[l]);
                mxv[l] = imax2(mxv[l], hh);
                H1[(size_t)j * P + l] = hh;
                e = sat_sub_u(e, e_del); t = sat_sub_u(hh, oe_del);
                E[(size_t)j * P + l] = imax2(e, t);
                f[l] = imax2(sat_sub_u(f[l], e_ins), sat_sub_u(hh, oe_ins));
                h[l] = H0[(size_t)j * P + l];
            }
        }
        /* lazy F (mimics SWPS3): H updated here cannot exceed the row maximum already taken */
        for (k = 0; k < 16 && !stop; ++k) {
            for (l = P - 1; l > 0; --l) f[l] = f[l - 1];
            f[0] = 0;
            for (j = 0; j < slen; ++j) {
                int any = 0;
                for (l = 0; l < P; ++l) {
                    int hh = imax2(H1[(size_t)j * P + l], f[l]);
                    H1[(size_t)j * P + l] = hh;
                    hh = sat_sub_u(hh, oe_ins);
                    f[l] = sat_sub_u(f[l], e_ins);
                    if (f[l] > hh) any = 1;
                }
                if (!any) { stop = 1; break; }
            }
        }
        for (l = 0; l < P; ++l) imax = imax2(imax, mxv[l]);
        if (imax >= minsc) {                                                   /* the b array of sub-optimal ends */
            if (n_b == 0 || (int32_t)b[n_b - 1] + 1 != i) {
                if (n_b == m_b) { m_b = m_b ? m_b << 1 : 8; b = (uint64_t *)realloc(b, 8 * (size_t)m_b); }
                b[n_b++] = (uint64_t)imax << 32 | (uint32_t)i;
            } else if ((int)(b[n_b - 1] >> 32) < imax) b[n_b - 1] = (uint64_t)imax << 32 | (uint32_t)i;
        }
        if (imax > gmax) {
            gmax = imax; te = i;
            memcpy(Hmax, H1, nv * sizeof(int));
            if (size == 1 ? (gmax + shift >= 255 || gmax >= endsc) : (gmax >= endsc)) break;
        }
        { int *t = H1; H1 = H0; H0 = t; }
    }
    r.score = size == 1 ? (gmax + shift < 255 ? gmax : 255) : gmax;
    r.te = te;
    if (size != 1 || r.score != 255) {                                         /* qe and the second best score */
        int best = -1, low, high;
        for (i = 0; i < slen * P; ++i)
            if (Hmax[i] > best) { best = Hmax[i]; r.qe = i / P + i % P * slen; }
        if (b) {
            i = (r.score + mx - 1) / mx;
            low = te - i; high = te + i;
            for (i = 0; i < n_b; ++i) {
                const int e = (int32_t)b[i];
                if ((e < low || e > high) && (int)(b[i] >> 32) > r.score2) { r.score2 = (int)(b[i] >> 32); r.te2 = e; }
            }
        }
    }
    free(b); free(qp); free(H0); free(H1); free(E); free(Hmax);
    return r;
}

/* bwa's ksw_align2 (the kswq_t cache argument dropped: the profile is rebuilt per call).  out[0..6] = score, te, qe,
 * score2, te2, tb, qb. */
void ksw_align2_ref(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat,
                    int o_del, int e_del, int o_ins, int e_ins, int xtra, int32_t *out, uint64_t *cells)
{
    const int size = (xtra & A_XBYTE) ? 1 : 2;
    kswr_ref_t r = ksw_vec_ref(size, qlen, query, tlen, target, m, mat, o_del, e_del, o_ins, e_ins, xtra, cells), rr;
    if (!((xtra & A_XSTART) == 0 || ((xtra & A_XSUBO) && r.score < (xtra & 0xffff)))) {
        uint8_t *q2 = (uint8_t *)malloc((size_t)(r.qe + 1 > 0 ? r.qe + 1 : 1)), *t2 = (uint8_t *)malloc((size_t)(tlen > 0 ? tlen : 1));
        int i;
        for (i = 0; i <= r.qe; ++i) q2[i] = query[r.qe - i];                   /* revseq(r.qe + 1, query) */
        if (tlen > 0) memcpy(t2, target, (size_t)tlen);
        for (i = 0; i <= r.te; ++i) t2[i] = target[r.te - i];                  /* revseq(r.te + 1, target) */
        rr = ksw_vec_ref(size, r.qe + 1, q2, tlen, t2, m, mat, o_del, e_del, o_ins, e_ins, A_XSTOP | r.score, cells);
        free(q2); free(t2);
        if (r.score == rr.score) { r.tb = r.te - rr.te; r.qb = r.qe - rr.qe; }
    }
    out[0] = r.score; out[1] = r.te; out[2] = r.qe; out[3] = r.score2; out[4] = r.te2; out[5] = r.tb; out[6] = r.qb;
}

typedef struct {
    const int8_t *mat; int o_del, e_del, o_ins, e_ins;
    const uint8_t *const *query, *const *target; const int32_t *qlen, *tlen, *xtra;
    int32_t *out; size_t lo, hi; uint64_t cells;
} align_job;

static void *align_worker(void *arg)
{
    align_job *j = (align_job *)arg;
    for (size_t i = j->lo; i < j->hi; ++i)
        ksw_align2_ref(j->qlen[i], j->query[i], j->tlen[i], j->target[i], 5, j->mat, j->o_del, j->e_del, j->o_ins, j->e_ins, j->xtra[i],
                       j->out + 7 * i, &j->cells);
    return 0;
}

/* n alignments (m = 5), out[7 i ..] as ksw_align2_ref; returns the DP cells evaluated (both passes) */
#include <pthread.h>
uint64_t ksw_align2_batch_ref(const int8_t *mat, int o_del, int e_del, int o_ins, int e_ins, const uint8_t *const *query,
                              const uint8_t *const *target, const int32_t *qlen, const int32_t *tlen, const int32_t *xtra, size_t n,
                              int32_t *out, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 64) nthreads = 64;
    align_job jobs[64];
    pthread_t th[64];
    const size_t per = (n + (size_t)nthreads - 1) / (size_t)nthreads;
    uint64_t cells = 0;
    for (int t = 0; t < nthreads; ++t) {
        align_job j = {mat, o_del, e_del, o_ins, e_ins, query, target, qlen, tlen, xtra, out, per * (size_t)t, 0, 0};
        j.hi = j.lo + per < n ? j.lo + per : n;
        if (j.lo > n) j.lo = n;
        jobs[t] = j;
        pthread_create(&th[t], 0, align_worker, &jobs[t]);
    }
    for (int t = 0; t < nthreads; ++t) { pthread_join(th[t], 0); cells += jobs[t].cells; }
    return cells;
}
