/*
 * ksw_global_ref.c — CPU ORACLE (test infrastructure, NOT product code) for SURVEY.md §8f row F4:
 * bwa's banded global alignment with CIGAR, ksw_global2 (and ksw_global = equal ins/del penalties).
 *
 * PARITY UNPINNED: ksw_global lives in the host software of the reference (peterpengwei/bwa-mem-quickassist,
 * bwa-0.7.8/ksw.c, named at /root/reference/README.md:7-18), which is not in this image; the RTL does not
 * implement it.  This is a restatement of the published algorithm (bwa ksw.c: eh[] row of {H(i-1,j-1), E(i,j)},
 * one direction byte per cell h | e<<2 | f<<4, backtrack from the last cell), pinned by analytic KATs, an
 * independent numpy global DP (oracle/py/full_dp.py) and CIGAR re-scoring in tests/test_oracle_global.py.
 */
#include "ksw_extend_ref.h"

#include <stdlib.h>
#include <string.h>

#define G_MINUS_INF (-0x40000000)

typedef struct { int32_t h, e; } geh_t;

static uint32_t *g_push(int *n, int *m, uint32_t *cigar, uint32_t op, int len)
{
    if (*n == 0 || op != (cigar[*n - 1] & 0xf)) {
        if (*n == *m) {
            *m = *m ? *m << 1 : 4;
            cigar = (uint32_t *)realloc(cigar, (size_t)*m * 4);
        }
        cigar[(*n)++] = (uint32_t)len << 4 | op;
    } else cigar[*n - 1] += (uint32_t)len << 4;
    return cigar;
}

/* returns the global score; if n_cigar_ and cigar_ are non-NULL, *cigar_ is malloc'ed (BAM encoding len<<4|op,
 * op 0 = M, 1 = I, 2 = D) and *n_cigar_ its length.  cells (optional) += DP cells evaluated. */
int ksw_global2_ref(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat,
                    int o_del, int e_del, int o_ins, int e_ins, int w, int *n_cigar_, uint32_t **cigar_, uint64_t *cells_)
{
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    geh_t *eh;
    uint8_t *z;
    int i, j, k, score, n_col;
    uint64_t cells = 0;
    if (n_cigar_) *n_cigar_ = 0;
    n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;                   /* columns of the backtrack matrix */
    z = (n_cigar_ && cigar_) ? (uint8_t *)malloc((size_t)(n_col > 0 ? n_col : 1) * (size_t)(tlen > 0 ? tlen : 1)) : 0;
    eh = (geh_t *)calloc((size_t)qlen + 2, sizeof(geh_t));
    /* first row: a leading insertion of j query bases, inside the band only */
    eh[0].h = 0; eh[0].e = G_MINUS_INF;
    for (j = 1; j <= qlen && j <= w; ++j) { eh[j].h = -(o_ins + e_ins * j); eh[j].e = G_MINUS_INF; }
    for (; j <= qlen; ++j) eh[j].h = eh[j].e = G_MINUS_INF;
    for (i = 0; i < tlen; ++i) {                                    /* target in the outer loop */
        int32_t f = G_MINUS_INF, h1, beg, end, t;
        const int8_t *srow = &mat[target[i] * m];
        uint8_t *zi = z ? &z[(size_t)i * (size_t)n_col] : 0;
        beg = i > w ? i - w : 0;
        end = i + w + 1 < qlen ? i + w + 1 : qlen;
        h1 = beg == 0 ? -(o_del + e_del * (i + 1)) : G_MINUS_INF;
        if (end > beg) cells += (uint64_t)(end - beg);
        for (j = beg; j < end; ++j) {
            /* eh[j] = {H(i-1,j-1), E(i,j)}, f = F(i,j), h1 = H(i,j-1);  M is kept apart from H so that the
             * direction bits are exact: H = max(M,E,F), E' = max(M-gapo, E) - gape, F' likewise */
            geh_t *p = &eh[j];
            int32_t h, M = p->h, e = p->e;
            uint8_t d;
            p->h = h1;
            M += srow[query[j]];
            d = M >= e ? 0 : 1;
            h = M >= e ? M : e;
            d = h >= f ? d : 2;
            h = h >= f ? h : f;
            h1 = h;
            t = M - oe_del;
            e -= e_del;
            d |= e > t ? 1 << 2 : 0;
            e = e > t ? e : t;
            p->e = e;
            t = M - oe_ins;
            f -= e_ins;
            d |= f > t ? 2 << 4 : 0;
            f = f > t ? f : t;
            if (zi) zi[j - beg] = d;
        }
        eh[end].h = h1; eh[end].e = G_MINUS_INF;
    }
    score = eh[qlen].h;
    if (z) {                                                        /* backtrack */
        int n_cigar = 0, m_cigar = 0, which = 0;
        uint32_t *cigar = 0, tmp;
        i = tlen - 1; k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1;   /* (i,k) = the last cell */
        while (i >= 0 && k >= 0) {
            which = z[(size_t)i * (size_t)n_col + (size_t)(k - (i > w ? i - w : 0))] >> (which << 1) & 3;
            if (which == 0) { cigar = g_push(&n_cigar, &m_cigar, cigar, 0, 1); --i; --k; }
            else if (which == 1) { cigar = g_push(&n_cigar, &m_cigar, cigar, 2, 1); --i; }
            else { cigar = g_push(&n_cigar, &m_cigar, cigar, 1, 1); --k; }
        }
        if (i >= 0) cigar = g_push(&n_cigar, &m_cigar, cigar, 2, i + 1);
        if (k >= 0) cigar = g_push(&n_cigar, &m_cigar, cigar, 1, k + 1);
        for (i = 0; i < n_cigar >> 1; ++i) { tmp = cigar[i]; cigar[i] = cigar[n_cigar - 1 - i]; cigar[n_cigar - 1 - i] = tmp; }
        *n_cigar_ = n_cigar; *cigar_ = cigar;
    }
    free(eh); free(z);
    if (cells_) *cells_ += cells;
    return score;
}
