/*
 * ksw_extend_ref.c — CPU ORACLE (test infrastructure, NOT product code).
 * See ksw_extend_ref.h for provenance and the "parity unpinned" statement.
 *
 * Every block cites the reference RTL lines it restates
 * (paths relative to the reference tree).
 */
#include "ksw_extend_ref.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int32_t h, e; } eh_t;

/* Per-thread scratch so the baseline is not dominated by malloc (bwa itself
 * callocs per call; reusing the row makes this a *stronger* CPU baseline). */
static __thread eh_t *tl_eh = NULL;
static __thread int   tl_eh_cap = 0;

static eh_t *eh_scratch(int qlen)
{
    if (tl_eh_cap < qlen + 2) {
        free(tl_eh);
        tl_eh_cap = qlen + 2 + 64;
        tl_eh = (eh_t *)malloc((size_t)tl_eh_cap * sizeof(eh_t));
    }
    return tl_eh;
}

/* wlim > 0: the band is clamped by this host-supplied limit (the RTL's H5/H6 words,
 * sw_pe_array_proc_element.v:925,933 -> sw_extend.v:1881,1890) instead of max_ins/max_del
 * computed from the scoring parameters. */
static int extend2_core(int qlen, const uint8_t *query, int tlen, const uint8_t *target,
                        int m, const int8_t *mat, int o_del, int e_del, int o_ins, int e_ins,
                        int w, int end_bonus, int zdrop, int h0,
                        int *qle_, int *tle_, int *gtle_, int *gscore_, int *max_off_,
                        int variant, uint64_t *cells_, int wlim)
{
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    eh_t *eh = eh_scratch(qlen);
    int i, j, k, beg, end, max, max_i, max_j, max_ie, gscore, max_off, max_ins, max_del;
    uint64_t cells = 0;

    /* K2 first row (sw_pe_array_sw_extend.v:1979,1957,1974,1818-1821,857-859):
     * eh[0].h=h0, eh[1].h=max(h0-oe_ins,0), then decay by e_ins while positive;
     * every e and every remaining h is 0. */
    memset(eh, 0, (size_t)(qlen + 1) * sizeof(eh_t));
    eh[0].h = h0;
    if (qlen >= 1) eh[1].h = h0 > oe_ins ? h0 - oe_ins : 0;
    for (j = 2; j <= qlen && eh[j - 1].h > e_ins; ++j) eh[j].h = eh[j - 1].h - e_ins;

    /* band clamp by the longest useful gap (P1; host-precomputed as H5/H6 in the
     * RTL: sw_pe_array_proc_element.v:925,933; applied at sw_extend.v:1881,1890) */
    for (i = 0, max = 0, k = m * m; i < k; ++i) max = max > mat[i] ? max : mat[i];
    max_ins = (int)((double)(qlen * max + end_bonus - o_ins) / e_ins + 1.);
    if (max_ins < 1) max_ins = 1;
    max_del = (int)((double)(qlen * max + end_bonus - o_del) / e_del + 1.);
    if (max_del < 1) max_del = 1;
    if (wlim > 0) max_ins = max_del = wlim;
    if (w > max_ins) w = max_ins;
    if (w > max_del) w = max_del;

    /* st1 init (:889,919,1009,929) */
    max = h0; max_i = max_j = -1; max_ie = -1; gscore = -1; max_off = 0;
    beg = 0; end = qlen;

    for (i = 0; i < tlen; ++i) {                         /* st4 row head (:1891) */
        int f = 0, h1, mrow = 0, mj = -1;
        const int8_t *srow = &mat[target[i] * m];        /* K6 (:1915-1940,1956) */
        /* K3 band clamp (:1803,1894-1897,1842,1898) */
        if (beg < i - w) beg = i - w;
        if (end > i + w + 1) end = i + w + 1;
        if (end > qlen) end = qlen;
        /* K4 column 0 (:1795-1796,1835); CPU semantics: only when beg==0 (quirk Q4 avoided) */
        if (beg == 0) {
            h1 = h0 - (o_del + e_del * (i + 1));
            if (h1 < 0) h1 = 0;
        } else h1 = 0;
        if (end > beg) cells += (uint64_t)(end - beg);
        for (j = beg; j < end; ++j) {                    /* K5 inner cell, pp0 (:1901) */
            eh_t *p = &eh[j];
            int h = p->h, e = p->e, s = srow[query[j]], t, base;
            p->h = h1;                                   /* eh[j].h <- H(i,j-1) (:1776) */
            if (variant == BSW_VARIANT_M) {
                int M = h ? h + s : 0;                   /* bwa>=0.7.9 */
                h = M > e ? M : e;
                h = h > f ? h : f;
                base = M;
            } else {
                h += s;                                  /* variant H: no zero test (:1797) */
                h = h > e ? h : e;                       /* (:1798) */
                h = h > f ? h : f;                       /* (:1809,1944) */
                base = h;                                /* t derives from h (:1866,1863) */
            }
            h1 = h;
            mj = mrow > h ? mj : j;                      /* ties -> later j (:1808,1816) */
            mrow = mrow > h ? mrow : h;
            t = base - oe_del; if (t < 0) t = 0;         /* (:1866) */
            e -= e_del; if (e < t) e = t;                /* (:1770-1771) */
            p->e = e;
            t = base - oe_ins; if (t < 0) t = 0;         /* (:1863,1865) */
            f -= e_ins; if (f < t) f = t;                /* (:1780-1781) */
        }
        eh[end].h = h1; eh[end].e = 0;                   /* K7 row tail (:1775) */
        if (j == qlen) {                                 /* (:1913,1941,1829-1833) ties -> later i */
            max_ie = gscore > h1 ? max_ie : i;
            gscore = gscore > h1 ? gscore : h1;
        }
        if (mrow == 0) break;                            /* (:1942) */
        if (mrow > max) {                                /* (:1959,1810,1845,1812-1813) */
            int off = mj - i; if (off < 0) off = -off;
            max = mrow; max_i = i; max_j = mj;
            if (off > max_off) max_off = off;
        } else if (zdrop > 0) {                          /* not in the RTL (Q3); C ABI requires it */
            if (i - max_i > mj - max_j) {
                if (max - mrow - ((i - max_i) - (mj - max_j)) * e_del > zdrop) break;
            } else {
                if (max - mrow - ((mj - max_j) - (i - max_i)) * e_ins > zdrop) break;
            }
        }
        /* K8 next-row range, CPU semantics (quirk Q5 avoided) */
        for (j = beg; j < end && eh[j].h == 0 && eh[j].e == 0; ++j) {}
        beg = j;
        for (j = end; j >= beg && eh[j].h == 0 && eh[j].e == 0; --j) {}
        end = j + 2 < qlen ? j + 2 : qlen;
    }
    /* K9 epilogue (:1841,1868,1794,1792,1815,1954) */
    if (qle_) *qle_ = max_j + 1;
    if (tle_) *tle_ = max_i + 1;
    if (gtle_) *gtle_ = max_ie + 1;
    if (gscore_) *gscore_ = gscore;
    if (max_off_) *max_off_ = max_off;
    if (cells_) *cells_ += cells;
    return max;
}

int ksw_extend2_ref(int qlen, const uint8_t *query, int tlen, const uint8_t *target,
                    int m, const int8_t *mat, int o_del, int e_del, int o_ins, int e_ins,
                    int w, int end_bonus, int zdrop, int h0,
                    int *qle_, int *tle_, int *gtle_, int *gscore_, int *max_off_,
                    int variant, uint64_t *cells_)
{
    return extend2_core(qlen, query, tlen, target, m, mat, o_del, e_del, o_ins, e_ins, w, end_bonus, zdrop, h0,
                        qle_, tle_, gtle_, gscore_, max_off_, variant, cells_, 0);
}

/* the same with the RTL's H5/H6 band limit in place of the formula (wlim > 0), as side_ref calls it */
int ksw_extend2_wlim_ref(int qlen, const uint8_t *query, int tlen, const uint8_t *target,
                         int m, const int8_t *mat, int o_del, int e_del, int o_ins, int e_ins,
                         int w, int end_bonus, int zdrop, int h0,
                         int *qle_, int *tle_, int *gtle_, int *gscore_, int *max_off_,
                         int variant, uint64_t *cells_, int wlim)
{
    return extend2_core(qlen, query, tlen, target, m, mat, o_del, e_del, o_ins, e_ins, w, end_bonus, zdrop, h0,
                        qle_, tle_, gtle_, gscore_, max_off_, variant, cells_, wlim);
}

int ksw_extend_ref(int qlen, const uint8_t *query, int tlen, const uint8_t *target,
                   int m, const int8_t *mat, int gapo, int gape,
                   int w, int end_bonus, int zdrop, int h0,
                   int *qle, int *tle, int *gtle, int *gscore, int *max_off,
                   int variant, uint64_t *cells)
{
    return ksw_extend2_ref(qlen, query, tlen, target, m, mat, gapo, gape, gapo, gape,
                           w, end_bonus, zdrop, h0, qle, tle, gtle, gscore, max_off, variant, cells);
}

/* One side with the MAX_BAND_TRY loop (P1: sw_pe_array_sw_extend.v:1963,1878,1837,
 * 1859,1822,1969-1970): each pass starts from fresh state (quirk Q6 avoided). */
static int side_ref(const bsw_params *p, int qlen, const uint8_t *q, int tlen, const uint8_t *t,
                    int end_bonus, int h0, int prev_score, int wlim, bsw_ext *x)
{
    int k, score = prev_score, tries = p->max_band_try > 0 ? p->max_band_try : 1;
    uint64_t cells = 0;
    memset(x, 0, sizeof(*x));
    for (k = 0; k < tries; ++k) {
        int prev = score, aw = p->w << k;
        score = extend2_core(qlen, q, tlen, t, 5, p->mat, p->o_del, p->e_del, p->o_ins, p->e_ins,
                             aw, end_bonus, p->zdrop, h0,
                             &x->qle, &x->tle, &x->gtle, &x->gscore, &x->max_off, p->variant, &cells, wlim);
        x->aw = aw;
        if (score == prev || x->max_off < (aw >> 1) + (aw >> 2)) break;
    }
    x->score = score;
    x->cells = (uint32_t)cells;
    return score;
}

void bsw_pair_ref(const bsw_params *p, const bsw_task *t, bsw_result *r)
{
    int score = t->init_score, sc0;
    memset(r, 0, sizeof(*r));
    r->tag = t->tag;
    r->left.aw = r->right.aw = p->w;                      /* a->w = aw[0] = aw[1] = opt->w */
    /* P2 left (sw_pe_array_proc_element.v:1670-1675,1666-1667,1630,1640-1641) */
    if (t->lqlen > 0) {
        score = side_ref(p, t->lqlen, t->lquery, t->ltlen, t->ltarget, p->pen_clip5, t->h0, score, t->wlim_l, &r->left);
        if (r->left.gscore <= 0 || r->left.gscore <= score - p->pen_clip5) {
            r->qb = t->qbeg - r->left.qle; r->rb = -r->left.tle; r->truesc = score;
        } else {
            r->qb = 0; r->rb = -r->left.gtle; r->truesc = r->left.gscore;
        }
    } else {
        score = r->truesc = t->h0; r->qb = 0; r->rb = 0;
    }
    /* P2 right: h0 = score after the left extension (:1671) */
    sc0 = score;
    if (t->rqlen > 0) {
        score = side_ref(p, t->rqlen, t->rquery, t->rtlen, t->rtarget, p->pen_clip3, sc0, score, t->wlim_r, &r->right);
        if (r->right.gscore <= 0 || r->right.gscore <= score - p->pen_clip3) {
            r->qe = r->right.qle; r->re = r->right.tle; r->truesc += score - sc0;
        } else {
            r->qe = t->rqlen; r->re = r->right.gtle; r->truesc += r->right.gscore - sc0;
        }
    } else {
        r->qe = 0; r->re = 0;
    }
    r->score = score;
    r->w = r->left.aw > r->right.aw ? r->left.aw : r->right.aw;   /* P3 (:1684,1669) */
}

/* ---- batch drivers ------------------------------------------------------- */
typedef struct {
    const bsw_params *p; const bsw_task *tasks; const bsw_ext_task *etasks;
    size_t n; bsw_result *out; bsw_ext *eout; size_t *next; pthread_mutex_t *mu;
} job_t;

#define CHUNK 256

static void run_range(job_t *jb, size_t lo, size_t hi)
{
    size_t i;
    if (jb->tasks) {
        for (i = lo; i < hi; ++i) bsw_pair_ref(jb->p, &jb->tasks[i], &jb->out[i]);
    } else {
        for (i = lo; i < hi; ++i) {
            const bsw_ext_task *t = &jb->etasks[i];
            bsw_ext *x = &jb->eout[i];
            uint64_t cells = 0;
            memset(x, 0, sizeof(*x));
            x->score = ksw_extend2_ref(t->qlen, t->query, t->tlen, t->target, 5, jb->p->mat,
                                       jb->p->o_del, jb->p->e_del, jb->p->o_ins, jb->p->e_ins,
                                       t->w, t->end_bonus, jb->p->zdrop, t->h0,
                                       &x->qle, &x->tle, &x->gtle, &x->gscore, &x->max_off,
                                       jb->p->variant, &cells);
            x->aw = t->w; x->cells = (uint32_t)cells;
        }
    }
}

static void *worker(void *arg)
{
    job_t *jb = (job_t *)arg;
    for (;;) {
        size_t lo, hi;
        pthread_mutex_lock(jb->mu);
        lo = *jb->next; *jb->next = lo + CHUNK;
        pthread_mutex_unlock(jb->mu);
        if (lo >= jb->n) break;
        hi = lo + CHUNK < jb->n ? lo + CHUNK : jb->n;
        run_range(jb, lo, hi);
    }
    free(tl_eh); tl_eh = NULL; tl_eh_cap = 0;
    return NULL;
}

static void run_batch(job_t *jb, int nthreads)
{
    size_t next = 0;
    pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
    jb->next = &next; jb->mu = &mu;
    if (nthreads <= 1) { run_range(jb, 0, jb->n); return; }
    {
        pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
        int i;
        for (i = 0; i < nthreads; ++i) pthread_create(&th[i], NULL, worker, jb);
        for (i = 0; i < nthreads; ++i) pthread_join(th[i], NULL);
        free(th);
    }
}

void bsw_pair_batch_ref(const bsw_params *p, const bsw_task *tasks, size_t n, bsw_result *out, int nthreads)
{
    job_t jb; memset(&jb, 0, sizeof(jb));
    jb.p = p; jb.tasks = tasks; jb.n = n; jb.out = out;
    run_batch(&jb, nthreads);
}

void bsw_ext_batch_ref(const bsw_params *p, const bsw_ext_task *tasks, size_t n, bsw_ext *out, int nthreads)
{
    job_t jb; memset(&jb, 0, sizeof(jb));
    jb.p = p; jb.etasks = tasks; jb.n = n; jb.eout = out;
    run_batch(&jb, nthreads);
}
