/*
 * bsw_glue.c — mem_chain2aln caller glue (host, plain C): the steps on either side of the hot path
 * (SURVEY.md §8f F2).  The reference host (bwa-0.7.8 + AALSDK) is not in the tree; what it must put
 * into a task header and what it reads back are pinned by the RTL's header/record fields
 * (sw_pe_array_proc_element.v:807-933 H0-H7, :1662-1665 R0-R4): left sequences arrive already
 * reversed (the PE walks both sides with ascending addresses, sw_pe_array_sw_extend.v:1844,1982),
 * rb/re/qe come back relative to the seed's ends.
 */
#include "../../include/bwa_sw_mi355.h"

#include <string.h>

int bsw_cal_max_gap(const bsw_params *p, int qlen)
{
    const int a = p->mat[0];
    int l_del = (int)((double)(qlen * a - p->o_del) / p->e_del + 1.);
    int l_ins = (int)((double)(qlen * a - p->o_ins) / p->e_ins + 1.);
    int l = l_del > l_ins ? l_del : l_ins;
    if (l < 1) l = 1;
    return l < (p->w << 1) ? l : (p->w << 1);
}

int bsw_chain_window(const bsw_params *p, const bsw_seed *seeds, int n, int l_query, int64_t l_pac, int64_t rmax[2])
{
    int i;
    if (!p || !seeds || n < 1 || !rmax || p->e_del < 1 || p->e_ins < 1) return BSW_E_INVAL;
    rmax[0] = l_pac << 1; rmax[1] = 0;
    for (i = 0; i < n; ++i) {
        const bsw_seed *t = &seeds[i];
        const int rq = l_query - t->qbeg - t->len;
        const int64_t b = t->rbeg - (t->qbeg + bsw_cal_max_gap(p, t->qbeg));
        const int64_t e = t->rbeg + t->len + (rq + bsw_cal_max_gap(p, rq));
        if (b < rmax[0]) rmax[0] = b;
        if (e > rmax[1]) rmax[1] = e;
    }
    if (rmax[0] < 0) rmax[0] = 0;
    if (rmax[1] > (l_pac << 1)) rmax[1] = l_pac << 1;
    if (rmax[0] < l_pac && l_pac < rmax[1]) {       /* never bridge the forward/reverse boundary */
        if (seeds[0].rbeg < l_pac) rmax[1] = l_pac; else rmax[0] = l_pac;
    }
    return BSW_OK;
}

size_t bsw_seed_scratch_bytes(const bsw_seed *s, int64_t rmax0)
{
    const int64_t lt = s->rbeg - rmax0;
    return (size_t)(s->qbeg > 0 ? s->qbeg : 0) + (size_t)(lt > 0 ? lt : 0);
}

int bsw_seed_to_task(const bsw_params *p, const bsw_seed *s, int l_query, const uint8_t *query,
                     int64_t rmax0, int64_t rmax1, const uint8_t *rseq,
                     uint8_t *scratch, size_t scratch_len, uint32_t tag, bsw_task *t)
{
    int i;
    int64_t lt, re, rt;
    if (!p || !s || !query || !rseq || !t) return BSW_E_INVAL;
    if (s->qbeg < 0 || s->len < 1 || s->qbeg + s->len > l_query) return BSW_E_INVAL;
    lt = s->rbeg - rmax0;
    re = s->rbeg + s->len - rmax0;
    rt = rmax1 - rmax0 - re;
    if (lt < 0 || re < 0 || rt < 0 || lt > BSW_MAX_TLEN || rt > BSW_MAX_TLEN) return BSW_E_LIMIT;
    memset(t, 0, sizeof(*t));
    if (s->qbeg > 0) {                               /* left extension: both sequences reversed */
        if (!scratch || scratch_len < (size_t)s->qbeg + (size_t)lt) return BSW_E_NOMEM;
        for (i = 0; i < s->qbeg; ++i) scratch[i] = query[s->qbeg - 1 - i];
        for (i = 0; i < (int)lt; ++i) scratch[s->qbeg + i] = rseq[lt - 1 - i];
        t->lquery = scratch; t->lqlen = s->qbeg;
        t->ltarget = scratch + s->qbeg; t->ltlen = (int32_t)lt;
    }
    if (s->qbeg + s->len != l_query) {               /* right extension */
        t->rquery = query + s->qbeg + s->len; t->rqlen = l_query - (s->qbeg + s->len);
        t->rtarget = rseq + re; t->rtlen = (int32_t)rt;
    }
    t->h0 = s->len * p->mat[0];                      /* s->len * opt->a */
    t->init_score = -1;                              /* a->score = -1 before the left extension */
    t->qbeg = s->qbeg;
    t->tag = tag;
    return BSW_OK;
}

int bsw_result_to_alnreg(const bsw_seed *s, const bsw_result *r, bsw_alnreg *a)
{
    if (!s || !r || !a) return BSW_E_INVAL;
    a->score = r->score; a->truesc = r->truesc; a->w = r->w;
    a->qb = r->qb;                        a->rb = s->rbeg + r->rb;            /* rb = -tle | -gtle | 0 */
    a->qe = s->qbeg + s->len + r->qe;     a->re = s->rbeg + s->len + r->re;
    return BSW_OK;
}

#define PAC_BASE(pac, l) ((pac)[(l) >> 2] >> ((~(l) & 3) << 1) & 3)

int64_t bsw_pac_get_seq(int64_t l_pac, const uint8_t *pac, int64_t beg, int64_t end, uint8_t *dst)
{
    int64_t k, l = 0;
    if (!pac || !dst) return BSW_E_INVAL;
    if (end < beg) { const int64_t x = beg; beg = end; end = x; }
    if (end > (l_pac << 1)) end = l_pac << 1;
    if (beg < 0) beg = 0;
    if (beg >= l_pac || end <= l_pac) {
        if (beg >= l_pac) {                          /* reverse strand: complement of the mirrored range */
            const int64_t beg_f = (l_pac << 1) - 1 - end, end_f = (l_pac << 1) - 1 - beg;
            for (k = end_f; k > beg_f; --k) dst[l++] = (uint8_t)(3 - PAC_BASE(pac, k));
        } else {
            for (k = beg; k < end; ++k) dst[l++] = (uint8_t)PAC_BASE(pac, k);
        }
    }
    return l;
}
