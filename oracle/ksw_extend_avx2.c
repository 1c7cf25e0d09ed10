/*
 * ksw_extend_avx2.c — STRONG CPU BASELINE (test infrastructure, NOT product code; see ksw_extend_ref.h).
 *
 * Inter-task SIMD ksw_extend2: 16 seeds per __m256i (int16 lanes), all walking DP row i together, each lane with
 * its own [beg,end) range, masks where a lane is outside its range — the way bwa-mem2 vectorises this path and the
 * CPU counterpart of the GPU lane kernels.  It follows the same algorithm as the scalar oracle (ksw_extend_ref.c:
 * K2-K9 of SURVEY.md §8a, sw_pe_array_sw_extend.v:1639-1705 with CPU semantics) and must return the same bytes:
 * tests/test_oracle_avx2.py compares every field of every seed, including the exact cell counts.
 * What does not fit the vector formulation (general 5x5 matrices, scores beyond int16, band retries) is handed to the
 * scalar oracle per seed, so the batch result is always complete.
 * "optional stronger baseline: inter-task AVX2 int16 version, clearly labelled as ours".
 */
#include "ksw_extend_ref.h"

#include <immintrin.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define VL 16                                   /* int16 lanes of a __m256i */

typedef struct {
    int16_t *H, *E;                             /* [column][lane] */
    int16_t *Q, *T;                             /* transposed query / target bases, [position][lane] */
    int cap_q, cap_t;
} scratch_t;

static void scratch_fit(scratch_t *s, int qmax, int tmax)
{
    if (s->cap_q < qmax + 2) {
        free(s->H); free(s->E); free(s->Q);
        s->cap_q = qmax + 2 + 64;
        s->H = (int16_t *)aligned_alloc(32, (size_t)s->cap_q * VL * sizeof(int16_t));
        s->E = (int16_t *)aligned_alloc(32, (size_t)s->cap_q * VL * sizeof(int16_t));
        s->Q = (int16_t *)aligned_alloc(32, (size_t)s->cap_q * VL * sizeof(int16_t));
    }
    if (s->cap_t < tmax + 1) {
        free(s->T);
        s->cap_t = tmax + 1 + 256;
        s->T = (int16_t *)aligned_alloc(32, (size_t)s->cap_t * VL * sizeof(int16_t));
    }
}

static void scratch_free(scratch_t *s) { free(s->H); free(s->E); free(s->Q); free(s->T); memset(s, 0, sizeof(*s)); }

typedef struct {                                /* one lane's ksw_extend2 call */
    int qlen, tlen, h0, w;
    const uint8_t *q, *t;
} lane_in;

/* bwa-style matrix: a on the diagonal of ACGT, -b off it, one score for any N */
static int bwa_style(const int8_t *mat, int *a, int *b, int *nsc)
{
    int i, j;
    *a = mat[0]; *b = -mat[1]; *nsc = mat[24];
    for (i = 0; i < 5; ++i)
        for (j = 0; j < 5; ++j) {
            const int want = (i == 4 || j == 4) ? *nsc : (i == j ? *a : -*b);
            if (mat[i * 5 + j] != want) return 0;
        }
    return 1;
}

/* One band try of up to 16 seeds.  in[l].w already holds min(w, max_ins, max_del).  Lanes l >= nl are idle. */
static void extend16(const bsw_params *p, int a, int b, int nsc, const lane_in *in, int nl, scratch_t *S, bsw_ext *out)
{
    const int oe_del = p->o_del + p->e_del, oe_ins = p->o_ins + p->e_ins, e_del = p->e_del, e_ins = p->e_ins;
    const int vm = p->variant == BSW_VARIANT_M, zdrop = p->zdrop;
    int qmax = 0, tmax = 0, l, i, j;
    int beg[VL], end[VL], mx[VL], max_i[VL], max_j[VL], max_ie[VL], gscore[VL], max_off[VL], alive[VL];
    uint32_t cells[VL];
    for (l = 0; l < nl; ++l) { if (in[l].qlen > qmax) qmax = in[l].qlen; if (in[l].tlen > tmax) tmax = in[l].tlen; }
    scratch_fit(S, qmax, tmax);
    int16_t *H = S->H, *E = S->E, *Q = S->Q, *T = S->T;
    memset(H, 0, (size_t)(qmax + 2) * VL * sizeof(int16_t));
    memset(E, 0, (size_t)(qmax + 2) * VL * sizeof(int16_t));
    memset(Q, 0, (size_t)(qmax + 1) * VL * sizeof(int16_t));
    memset(T, 0, (size_t)(tmax + 1) * VL * sizeof(int16_t));
    for (l = 0; l < VL; ++l) {
        beg[l] = end[l] = 0; mx[l] = 0; max_i[l] = max_j[l] = max_ie[l] = gscore[l] = -1; max_off[l] = 0; alive[l] = 0; cells[l] = 0;
    }
    for (l = 0; l < nl; ++l) {
        const lane_in *x = &in[l];
        for (j = 0; j < x->qlen; ++j) Q[j * VL + l] = x->q[j] > 4 ? 4 : x->q[j];
        for (i = 0; i < x->tlen; ++i) T[i * VL + l] = x->t[i] > 4 ? 4 : x->t[i];
        /* K2 first row (sw_pe_array_sw_extend.v:1979,1957,1974) */
        H[0 * VL + l] = (int16_t)x->h0;
        if (x->qlen >= 1) H[1 * VL + l] = (int16_t)(x->h0 > oe_ins ? x->h0 - oe_ins : 0);
        for (j = 2; j <= x->qlen && H[(j - 1) * VL + l] > e_ins; ++j) H[j * VL + l] = (int16_t)(H[(j - 1) * VL + l] - e_ins);
        end[l] = x->qlen; mx[l] = x->h0; alive[l] = x->tlen > 0;
    }
    const __m256i vA = _mm256_set1_epi16((short)a), vNB = _mm256_set1_epi16((short)-b), vN = _mm256_set1_epi16((short)nsc);
    const __m256i v3 = _mm256_set1_epi16(3), vZ = _mm256_setzero_si256();
    const __m256i vOED = _mm256_set1_epi16((short)oe_del), vOEI = _mm256_set1_epi16((short)oe_ins);
    const __m256i vED = _mm256_set1_epi16((short)e_del), vEI = _mm256_set1_epi16((short)e_ins);
    for (i = 0; i < tmax; ++i) {
        int16_t begv[VL] __attribute__((aligned(32))), endv[VL] __attribute__((aligned(32))), h1v[VL] __attribute__((aligned(32)));
        int16_t mv[VL] __attribute__((aligned(32))), mjv[VL] __attribute__((aligned(32)));
        int jlo = 1 << 30, jhi = -1, any = 0;
        for (l = 0; l < VL; ++l) {
            const int act = l < nl && alive[l] && i < in[l].tlen;
            if (!act) { begv[l] = 1; endv[l] = 0; h1v[l] = 0; continue; }      /* empty range: every mask below is off */
            /* K3 band clamp (:1803,1894-1897,1842,1898) */
            if (beg[l] < i - in[l].w) beg[l] = i - in[l].w;
            if (end[l] > i + in[l].w + 1) end[l] = i + in[l].w + 1;
            if (end[l] > in[l].qlen) end[l] = in[l].qlen;
            /* K4 column 0 (:1795-1796,1835), CPU semantics */
            int h1 = 0;
            if (beg[l] == 0) { h1 = in[l].h0 - (p->o_del + e_del * (i + 1)); if (h1 < 0) h1 = 0; }
            begv[l] = (int16_t)beg[l]; endv[l] = (int16_t)end[l]; h1v[l] = (int16_t)h1;
            if (end[l] > beg[l]) { cells[l] += (uint32_t)(end[l] - beg[l]); if (beg[l] < jlo) jlo = beg[l]; if (end[l] > jhi) jhi = end[l]; }
            any = 1;
        }
        if (!any) break;
        {
            const __m256i vbeg = _mm256_load_si256((const __m256i *)begv), vend = _mm256_load_si256((const __m256i *)endv);
            const __m256i vt = _mm256_load_si256((const __m256i *)&T[i * VL]);
            __m256i h1 = _mm256_load_si256((const __m256i *)h1v), f = vZ, m = vZ, mj = _mm256_set1_epi16(-1);
            for (j = jlo; j < jhi; ++j) {                /* K5 inner cell for the 16 lanes (:1797-1816,1863-1866) */
                const __m256i vj = _mm256_set1_epi16((short)j);
                /* in range: beg <= j < end */
                const __m256i in_r = _mm256_andnot_si256(_mm256_cmpgt_epi16(vbeg, vj), _mm256_cmpgt_epi16(vend, vj));
                __m256i h = _mm256_load_si256((const __m256i *)&H[j * VL]), e = _mm256_load_si256((const __m256i *)&E[j * VL]);
                const __m256i vq = _mm256_load_si256((const __m256i *)&Q[j * VL]);
                __m256i s = _mm256_blendv_epi8(vNB, vA, _mm256_cmpeq_epi16(vt, vq));
                s = _mm256_blendv_epi8(s, vN, _mm256_cmpgt_epi16(_mm256_max_epi16(vt, vq), v3));
                _mm256_store_si256((__m256i *)&H[j * VL], _mm256_blendv_epi8(h, h1, in_r));      /* eh[j].h <- H(i,j-1) (:1776) */
                __m256i base;
                if (vm) {
                    const __m256i M = _mm256_andnot_si256(_mm256_cmpeq_epi16(h, vZ), _mm256_add_epi16(h, s));   /* M = h ? h + s : 0 */
                    h = _mm256_max_epi16(_mm256_max_epi16(M, e), f);
                    base = M;
                } else {
                    h = _mm256_max_epi16(_mm256_max_epi16(_mm256_add_epi16(h, s), e), f);
                    base = h;
                }
                h1 = _mm256_blendv_epi8(h1, h, in_r);
                /* ties -> later j: mj = m > h ? mj : j */
                mj = _mm256_blendv_epi8(mj, vj, _mm256_andnot_si256(_mm256_cmpgt_epi16(m, h), in_r));
                m = _mm256_blendv_epi8(m, _mm256_max_epi16(m, h), in_r);
                __m256i t = _mm256_max_epi16(_mm256_sub_epi16(base, vOED), vZ);
                e = _mm256_max_epi16(_mm256_sub_epi16(e, vED), t);
                {
                    const __m256i e0 = _mm256_load_si256((const __m256i *)&E[j * VL]);
                    _mm256_store_si256((__m256i *)&E[j * VL], _mm256_blendv_epi8(e0, e, in_r));
                }
                t = _mm256_max_epi16(_mm256_sub_epi16(base, vOEI), vZ);
                f = _mm256_blendv_epi8(f, _mm256_max_epi16(_mm256_sub_epi16(f, vEI), t), in_r);
            }
            _mm256_store_si256((__m256i *)h1v, h1);
            _mm256_store_si256((__m256i *)mv, m);
            _mm256_store_si256((__m256i *)mjv, mj);
        }
        for (l = 0; l < nl; ++l) {                       /* K7 / K8 per lane */
            if (!(alive[l] && i < in[l].tlen)) continue;
            const int qlen = in[l].qlen, h1 = h1v[l], mrow = mv[l], mjl = mjv[l];
            const int jfin = beg[l] > end[l] ? beg[l] : end[l];
            H[end[l] * VL + l] = (int16_t)h1; E[end[l] * VL + l] = 0;         /* (:1775) */
            if (jfin == qlen) {                              /* ties -> later i (:1829-1833) */
                max_ie[l] = gscore[l] > h1 ? max_ie[l] : i;
                gscore[l] = gscore[l] > h1 ? gscore[l] : h1;
            }
            if (mrow == 0) { alive[l] = 0; continue; }       /* (:1942) */
            if (mrow > mx[l]) {
                int off = mjl - i; if (off < 0) off = -off;
                mx[l] = mrow; max_i[l] = i; max_j[l] = mjl;
                if (off > max_off[l]) max_off[l] = off;
            } else if (zdrop > 0) {
                if (i - max_i[l] > mjl - max_j[l]) {
                    if (mx[l] - mrow - ((i - max_i[l]) - (mjl - max_j[l])) * e_del > zdrop) { alive[l] = 0; continue; }
                } else {
                    if (mx[l] - mrow - ((mjl - max_j[l]) - (i - max_i[l])) * e_ins > zdrop) { alive[l] = 0; continue; }
                }
            }
            for (j = beg[l]; j < end[l] && H[j * VL + l] == 0 && E[j * VL + l] == 0; ++j) {}
            beg[l] = j;
            for (j = end[l]; j >= beg[l] && H[j * VL + l] == 0 && E[j * VL + l] == 0; --j) {}
            end[l] = j + 2 < qlen ? j + 2 : qlen;
        }
    }
    for (l = 0; l < nl; ++l) {
        bsw_ext *x = &out[l];
        memset(x, 0, sizeof(*x));
        x->score = mx[l]; x->qle = max_j[l] + 1; x->tle = max_i[l] + 1; x->gtle = max_ie[l] + 1;
        x->gscore = gscore[l]; x->max_off = max_off[l]; x->cells = cells[l];
    }
}

/* band limit exactly as extend2_core computes it */
static int band_of(const bsw_params *p, int mxs, int qlen, int end_bonus, int w, int wlim)
{
    int max_ins = (int)((double)(qlen * mxs + end_bonus - p->o_ins) / p->e_ins + 1.);
    int max_del = (int)((double)(qlen * mxs + end_bonus - p->o_del) / p->e_del + 1.);
    if (max_ins < 1) max_ins = 1;
    if (max_del < 1) max_del = 1;
    if (wlim > 0) max_ins = max_del = wlim;
    if (w > max_ins) w = max_ins;
    if (w > max_del) w = max_del;
    return w;
}

/* ---- pair driver over groups of 16 seeds (mem_chain2aln semantics of bsw_pair_ref) ---- */
typedef struct {
    const bsw_params *p; const bsw_task *tasks; size_t n; bsw_result *out; size_t *next; pthread_mutex_t *mu;
    int a, b, nsc, mxs, vec_ok;
} job_t;

#define CHUNK 512

static int cmp_desc(const void *x, const void *y)
{
    const uint64_t a = *(const uint64_t *)x, b = *(const uint64_t *)y;
    return a < b ? 1 : a > b ? -1 : 0;
}

/* one side of tasks[idx[0..cnt)]: first band try in vector groups, everything after it by the scalar oracle */
static void side_group(const job_t *jb, scratch_t *S, const uint32_t *idx, int cnt, int side)
{
    const bsw_params *p = jb->p;
    const int tries = p->max_band_try > 0 ? p->max_band_try : 1;
    int g, l;
    for (g = 0; g < cnt; g += VL) {
        const int nl = cnt - g < VL ? cnt - g : VL;
        lane_in in[VL];
        bsw_ext ex[VL];
        int prev[VL];
        for (l = 0; l < nl; ++l) {
            const bsw_task *t = &jb->tasks[idx[g + l]];
            bsw_result *r = &jb->out[idx[g + l]];
            if (side == 0) {
                in[l].qlen = t->lqlen; in[l].tlen = t->ltlen; in[l].q = t->lquery; in[l].t = t->ltarget; in[l].h0 = t->h0;
                in[l].w = band_of(p, jb->mxs, t->lqlen, p->pen_clip5, p->w, t->wlim_l);
                prev[l] = t->init_score;
            } else {
                in[l].qlen = t->rqlen; in[l].tlen = t->rtlen; in[l].q = t->rquery; in[l].t = t->rtarget; in[l].h0 = r->score;
                in[l].w = band_of(p, jb->mxs, t->rqlen, p->pen_clip3, p->w, t->wlim_r);
                prev[l] = r->score;
            }
        }
        extend16(p, jb->a, jb->b, jb->nsc, in, nl, S, ex);
        for (l = 0; l < nl; ++l) {
            const bsw_task *t = &jb->tasks[idx[g + l]];
            bsw_result *r = &jb->out[idx[g + l]];
            bsw_ext *x = side ? &r->right : &r->left;
            int score = ex[l].score, k;
            uint64_t cells = ex[l].cells;
            *x = ex[l];
            x->aw = p->w;
            /* MAX_BAND_TRY (P1): further passes start from fresh state, one seed at a time */
            for (k = 1; k < tries && !(score == prev[l] || x->max_off < ((p->w << (k - 1)) >> 1) + ((p->w << (k - 1)) >> 2)); ++k) {
                const int aw = p->w << k;
                prev[l] = score;
                score = ksw_extend2_wlim_ref(in[l].qlen, in[l].q, in[l].tlen, in[l].t, 5, p->mat, p->o_del, p->e_del, p->o_ins, p->e_ins,
                                             aw, side ? p->pen_clip3 : p->pen_clip5, p->zdrop, in[l].h0,
                                             &x->qle, &x->tle, &x->gtle, &x->gscore, &x->max_off, p->variant, &cells,
                                             side ? t->wlim_r : t->wlim_l);
                x->aw = aw;
            }
            x->score = score; x->cells = (uint32_t)cells;
            if (side == 0) {
                if (x->gscore <= 0 || x->gscore <= score - p->pen_clip5) { r->qb = t->qbeg - x->qle; r->rb = -x->tle; r->truesc = score; }
                else { r->qb = 0; r->rb = -x->gtle; r->truesc = x->gscore; }
                r->score = score;
            } else {
                const int sc0 = in[l].h0;
                if (x->gscore <= 0 || x->gscore <= score - p->pen_clip3) { r->qe = x->qle; r->re = x->tle; r->truesc += score - sc0; }
                else { r->qe = t->rqlen; r->re = x->gtle; r->truesc += x->gscore - sc0; }
                r->score = score;
            }
        }
    }
}

static void run_range(const job_t *jb, scratch_t *S, size_t lo, size_t hi)
{
    const bsw_params *p = jb->p;
    uint64_t key[CHUNK];
    uint32_t idx[CHUNK];
    size_t i;
    int cnt, side;
    if (!jb->vec_ok) { for (i = lo; i < hi; ++i) bsw_pair_ref(p, &jb->tasks[i], &jb->out[i]); return; }
    for (i = lo; i < hi; ++i) {                       /* seeds the int16 lanes cannot hold take the scalar oracle whole */
        const bsw_task *t = &jb->tasks[i];
        bsw_result *r = &jb->out[i];
        if ((long)t->h0 + (long)(t->lqlen + t->rqlen) * jb->mxs > 30000 || t->h0 <= 0 || t->ltlen > 32000 || t->rtlen > 32000 ||
            t->lqlen > 32000 || t->rqlen > 32000) { bsw_pair_ref(p, t, r); continue; }
        memset(r, 0, sizeof(*r));
        r->tag = t->tag;
        r->left.aw = r->right.aw = p->w;
        r->score = r->truesc = t->h0;                   /* a side with qlen == 0 is skipped, values kept */
    }
    for (side = 0; side < 2; ++side) {
        cnt = 0;
        for (i = lo; i < hi; ++i) {
            const bsw_task *t = &jb->tasks[i];
            const int ql = side ? t->rqlen : t->lqlen;
            if ((long)t->h0 + (long)(t->lqlen + t->rqlen) * jb->mxs > 30000 || t->h0 <= 0 || t->ltlen > 32000 || t->rtlen > 32000 ||
                t->lqlen > 32000 || t->rqlen > 32000) continue;
            if (ql > 0) key[cnt++] = ((uint64_t)ql << 32) | (uint32_t)i;
        }
        qsort(key, (size_t)cnt, sizeof(uint64_t), cmp_desc);         /* longest queries first: lanes of a group finish together */
        for (i = 0; i < (size_t)cnt; ++i) idx[i] = (uint32_t)key[i];
        side_group(jb, S, idx, cnt, side);
    }
    for (i = lo; i < hi; ++i) {
        bsw_result *r = &jb->out[i];
        r->w = r->left.aw > r->right.aw ? r->left.aw : r->right.aw;
    }
}

static void *worker(void *arg)
{
    job_t *jb = (job_t *)arg;
    scratch_t S; memset(&S, 0, sizeof(S));
    for (;;) {
        size_t lo, hi;
        pthread_mutex_lock(jb->mu);
        lo = *jb->next; *jb->next = lo + CHUNK;
        pthread_mutex_unlock(jb->mu);
        if (lo >= jb->n) break;
        hi = lo + CHUNK < jb->n ? lo + CHUNK : jb->n;
        run_range(jb, &S, lo, hi);
    }
    scratch_free(&S);
    return NULL;
}

/* Same contract and same bytes as bsw_pair_batch_ref. */
void bsw_pair_batch_avx2(const bsw_params *p, const bsw_task *tasks, size_t n, bsw_result *out, int nthreads)
{
    job_t jb; memset(&jb, 0, sizeof(jb));
    size_t next = 0;
    pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
    int i;
    jb.p = p; jb.tasks = tasks; jb.n = n; jb.out = out; jb.next = &next; jb.mu = &mu;
    for (i = 0, jb.mxs = 0; i < 25; ++i) jb.mxs = jb.mxs > p->mat[i] ? jb.mxs : p->mat[i];
    jb.vec_ok = bwa_style(p->mat, &jb.a, &jb.b, &jb.nsc) && p->o_del + p->e_del < 16000 && p->o_ins + p->e_ins < 16000 &&
                p->e_del >= 1 && p->e_ins >= 1 && jb.b < 16000 && jb.a > 0 && jb.a < 128;
    if (nthreads <= 1) { worker(&jb); return; }
    {
        pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
        for (i = 0; i < nthreads; ++i) pthread_create(&th[i], NULL, worker, &jb);
        for (i = 0; i < nthreads; ++i) pthread_join(th[i], NULL);
        free(th);
    }
}
