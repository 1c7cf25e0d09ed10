/*
 * rowsync_model.c — CPU MODEL of the row-synchronous formulation used by the
 * wave-per-task HIP kernel (test infrastructure, NOT product code).
 *
 * The HIP kernel (bwa-mem-sw_amd/csrc/bsw_wave_kernel.hip) cannot run in the build
 * container (no GPU), so the algebra it relies on is checked here against the
 * scalar oracle on millions of random tasks:
 *   - the intra-row F dependency rewritten as an exclusive prefix max of
 *     G_k = max(base_k - oe_ins, 0) + k*e_ins   (valid for o_ins >= 0),
 *   - row max / arg-max with "ties take the later j" as max over keys (h<<KB | j),
 *   - persistent eh[] with writes masked to [beg,end] (stale entries survive),
 *   - next-row range from the first / last non-zero (h|e) in [beg,end].
 * Restates the same reference units as oracle/ksw_extend_ref.c (K2-K9 of SURVEY.md §8a:
 * sw_pe_array_sw_extend.v:1763-1983).
 */
#include "ksw_extend_ref.h"

#include <stdlib.h>
#include <string.h>

#define KB 10                      /* key bits for the column index (BSW_MAX_QLEN < 1<<KB) */
#define NEG (-(1 << 29))

static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }

int ksw_extend2_rowsync_model(int qlen, const uint8_t *query, int tlen, const uint8_t *target,
                              int m, const int8_t *mat, int o_del, int e_del, int o_ins, int e_ins,
                              int w, int end_bonus, int zdrop, int h0,
                              int *qle_, int *tle_, int *gtle_, int *gscore_, int *max_off_,
                              int variant, uint64_t *cells_)
{
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    int *X = (int *)calloc((size_t)(qlen + 2) * 8, sizeof(int));
    int *E = X + (qlen + 2), *Mv = E + (qlen + 2), *ht = Mv + (qlen + 2), *G = ht + (qlen + 2);
    int *P = G + (qlen + 2), *hv = P + (qlen + 2), *Xn = hv + (qlen + 2);
    int i, j, k, beg = 0, end = qlen, max, max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0;
    int mx = 0, max_ins, max_del;
    uint64_t cells = 0;

    /* first row: closed form of the K2 recurrence, evaluated per column */
    X[0] = h0;
    for (j = 1; j <= qlen; ++j) X[j] = imax(h0 - oe_ins - (j - 1) * e_ins, 0);

    for (i = 0, k = m * m; i < k; ++i) mx = imax(mx, mat[i]);
    /* integer form of (int)((double)x/e + 1.) = trunc((x+e)/e) */
    max_ins = imax((qlen * mx + end_bonus - o_ins + e_ins) / e_ins, 1);
    max_del = imax((qlen * mx + end_bonus - o_del + e_del) / e_del, 1);
    w = imin(w, imin(max_ins, max_del));

    max = h0;
    for (i = 0; i < tlen; ++i) {
        int h1_init, hlast, jfin, mrow, mj, run;
        const int8_t *srow = &mat[target[i] * m];
        beg = imax(beg, i - w);
        end = imin(imin(end, i + w + 1), qlen);
        h1_init = beg == 0 ? imax(h0 - (o_del + e_del * (i + 1)), 0) : 0;
        if (end > beg) cells += (uint64_t)(end - beg);

        /* phase 1 (all columns independent) */
        for (j = beg; j < end; ++j) {
            int s = srow[query[j]], base;
            Mv[j] = variant == BSW_VARIANT_M ? (X[j] ? X[j] + s : 0) : X[j] + s;
            ht[j] = imax(Mv[j], E[j]);
            base = variant == BSW_VARIANT_M ? Mv[j] : ht[j];
            G[j] = imax(base - oe_ins, 0) + j * e_ins;
        }
        /* phase 2: exclusive prefix max (the wave scan) */
        run = NEG;
        for (j = beg; j < end; ++j) { P[j] = run; run = imax(run, G[j]); }
        /* phase 3 */
        {
            int mkey = -1;
            for (j = beg; j < end; ++j) {
                int f = imax(P[j] - (j - 1) * e_ins, 0), base, key;
                hv[j] = imax(ht[j], f);
                base = variant == BSW_VARIANT_M ? Mv[j] : hv[j];
                E[j] = imax(E[j] - e_del, imax(base - oe_del, 0));
                key = (hv[j] << KB) | j;
                mkey = imax(mkey, key);
            }
            if (mkey < 0) { mrow = 0; mj = -1; }
            else { mrow = mkey >> KB; mj = mkey & ((1 << KB) - 1); }
        }
        /* phase 4: shifted write-back of H, masked to [beg,end] */
        for (j = beg; j <= end; ++j) Xn[j] = j == beg ? h1_init : hv[j - 1];
        for (j = beg; j <= end; ++j) X[j] = Xn[j];
        E[end] = 0;
        hlast = end > beg ? hv[end - 1] : h1_init;
        jfin = imax(beg, end);

        if (jfin == qlen) {
            max_ie = gscore > hlast ? max_ie : i;
            gscore = imax(gscore, hlast);
        }
        if (mrow == 0) break;
        if (mrow > max) {
            max = mrow; max_i = i; max_j = mj;
            max_off = imax(max_off, abs(mj - i));
        } else if (zdrop > 0) {
            if (i - max_i > mj - max_j) {
                if (max - mrow - ((i - max_i) - (mj - max_j)) * e_del > zdrop) break;
            } else {
                if (max - mrow - ((mj - max_j) - (i - max_i)) * e_ins > zdrop) break;
            }
        }
        /* next-row range: first nz in [beg,end), last nz in [beg',end] */
        {
            int first = end, last;
            for (j = beg; j < end; ++j) if (X[j] | E[j]) { first = j; break; }
            last = first - 1;
            for (j = end; j >= first; --j) if (X[j] | E[j]) { last = j; break; }
            beg = first;
            end = imin(last + 2, qlen);
        }
    }
    if (qle_) *qle_ = max_j + 1;
    if (tle_) *tle_ = max_i + 1;
    if (gtle_) *gtle_ = max_ie + 1;
    if (gscore_) *gscore_ = gscore;
    if (max_off_) *max_off_ = max_off;
    if (cells_) *cells_ += cells;
    free(X);
    return max;
}
