/*
 * bsw_refbatch.c — codec for the reference accelerator's wire format (host, plain C).
 *
 * Task batch  : 65536 x 32-bit words = 256 KiB (bwa_mem_sw.v:163-166, tbb.v:59)
 *   W[0]  G0 = {e_ins[31:24], o_ins[23:16], e_del[15:8], o_del[7:0]}   (proc_element.v:816-819)
 *   W[1]  G1 = {-, w[23:16], pen_clip3[15:8], pen_clip5[7:0]}           (proc_element.v:916-917)
 *   W[2]  N  = number of tasks                                          (task_parse.v:942-944)
 *   W[8+8i .. 15+8i] header of task i                                   (task_parse.v:1931-1940)
 *     H0 {tlen0[26:16], qlen0[7:0]}  H1 {tlen1[26:16], qlen1[7:0]}
 *     H2 data position in words: pos = 8+8N - H2(0) + H2(i)             (task_parse.v:1924,1929)
 *     H3 {qbeg[31:16], initial score[15:0]}  H4 h0[7:0]                 (proc_element.v:872-873,827)
 *     H5 {max_del[31:16], max_ins[15:0]} left   H6 same, right          (proc_element.v:925,933)
 *     H7 tag                                                            (proc_element.v:807)
 *   data: ceil((ql0+ql1+tl0+tl1)/8) words, 8 bases per word, 4 bits each, first base in
 *         bits [31:28], one contiguous nibble stream leftQ,rightQ,leftT,rightT
 *         (proc_element.v:1638,1677,1682,1637,1683)
 * Result batch: 4096 words = 16 KiB (rbb.v:59): 5-word records
 *   R0 tag, R1 {qe[31:16], qb[15:0]}, R2 {re[31:16], rb[15:0]}, R3 {truesc[31:16], score[15:0]},
 *   R4 w                                                                (proc_element.v:1662-1665)
 * Field widths are the RTL's; a task that does not fit ends the batch.
 */
#include "../../include/bwa_sw_mi355.h"

#include <string.h>

static int gap_lim(int qlen, int mx, int end_bonus, int o, int e)
{
    int l = (qlen * mx + end_bonus - o + e) / e;
    return l < 1 ? 1 : (l > 32767 ? 32767 : l);
}

static int fits_u8(int v) { return v >= 0 && v <= 255; }

int bsw_refbatch_encode(const bsw_params *p, const bsw_task *tasks, size_t n, uint32_t *W)
{
    size_t i, ntask = 0, words = 0, k;
    uint32_t pos;
    if (!p || !W || (!tasks && n)) return BSW_E_INVAL;
    if (!fits_u8(p->o_del) || !fits_u8(p->e_del) || !fits_u8(p->o_ins) || !fits_u8(p->e_ins) ||
        !fits_u8(p->w) || !fits_u8(p->pen_clip5) || !fits_u8(p->pen_clip3)) return BSW_E_LIMIT;
    /* how many tasks fit: field widths, 819 results, 65536 words */
    for (i = 0; i < n && ntask < BSW_REFBATCH_MAX_TASKS; ++i) {
        const bsw_task *t = &tasks[i];
        size_t dw;
        if (t->lqlen < 0 || t->lqlen > 255 || t->rqlen < 0 || t->rqlen > 255 ||
            t->ltlen < 0 || t->ltlen > 2047 || t->rtlen < 0 || t->rtlen > 2047) break;
        if (t->h0 < 1 || t->h0 > 127) break;                          /* int8 datapath */
        if (t->qbeg < 0 || t->qbeg > 65535 || t->init_score < -32768 || t->init_score > 32767) break;
        if (t->lqlen + t->rqlen + t->ltlen + t->rtlen > 2048) break;  /* 2048 x 4b query_mem */
        dw = (size_t)(t->lqlen + t->rqlen + t->ltlen + t->rtlen + 7) / 8;
        if (8 + 8 * (ntask + 1) + words + dw > BSW_REFBATCH_IN_WORDS) break;
        words += dw;
        ++ntask;
    }
    memset(W, 0, BSW_REFBATCH_IN_WORDS * sizeof(uint32_t));
    W[0] = ((uint32_t)p->e_ins << 24) | ((uint32_t)p->o_ins << 16) | ((uint32_t)p->e_del << 8) | (uint32_t)p->o_del;
    W[1] = ((uint32_t)p->w << 16) | ((uint32_t)p->pen_clip3 << 8) | (uint32_t)p->pen_clip5;
    W[2] = (uint32_t)ntask;
    pos = (uint32_t)(8 + 8 * ntask);
    {
        int mx = 0;
        for (k = 0; k < 25; ++k) mx = mx > p->mat[k] ? mx : p->mat[k];
        for (i = 0; i < ntask; ++i) {
            const bsw_task *t = &tasks[i];
            uint32_t *H = &W[8 + 8 * i];
            const uint8_t *src[4];
            int len[4], s, nib = 0;
            H[0] = ((uint32_t)t->ltlen << 16) | (uint32_t)t->lqlen;
            H[1] = ((uint32_t)t->rtlen << 16) | (uint32_t)t->rqlen;
            H[2] = pos;
            H[3] = ((uint32_t)t->qbeg << 16) | ((uint32_t)t->init_score & 0xffffu);
            H[4] = (uint32_t)t->h0;
            /* H5/H6: a task that carries its own band limits (wlim > 0) ships them as max_ins = max_del */
            if (t->wlim_l > 0) { uint32_t l = (uint32_t)(t->wlim_l > 32767 ? 32767 : t->wlim_l); H[5] = (l << 16) | l; }
            else H[5] = ((uint32_t)gap_lim(t->lqlen, mx, p->pen_clip5, p->o_del, p->e_del) << 16) |
                        (uint32_t)gap_lim(t->lqlen, mx, p->pen_clip5, p->o_ins, p->e_ins);
            if (t->wlim_r > 0) { uint32_t l = (uint32_t)(t->wlim_r > 32767 ? 32767 : t->wlim_r); H[6] = (l << 16) | l; }
            else H[6] = ((uint32_t)gap_lim(t->rqlen, mx, p->pen_clip3, p->o_del, p->e_del) << 16) |
                        (uint32_t)gap_lim(t->rqlen, mx, p->pen_clip3, p->o_ins, p->e_ins);
            H[7] = t->tag;
            src[0] = t->lquery; len[0] = t->lqlen; src[1] = t->rquery; len[1] = t->rqlen;
            src[2] = t->ltarget; len[2] = t->ltlen; src[3] = t->rtarget; len[3] = t->rtlen;
            for (s = 0; s < 4; ++s) {
                int j;
                for (j = 0; j < len[s]; ++j, ++nib) {
                    uint32_t b = src[s][j] > 4 ? 4u : src[s][j];
                    W[pos + (uint32_t)(nib >> 3)] |= b << (28 - 4 * (nib & 7));   /* MSB-first */
                }
            }
            pos += (uint32_t)((nib + 7) >> 3);
        }
    }
    return (int)ntask;
}

int bsw_refbatch_decode(const uint32_t *W, bsw_params *p, bsw_task *tasks, size_t max_tasks,
                        uint8_t *seqbuf, size_t seqbuf_len)
{
    uint32_t n, i;
    size_t used = 0;
    int64_t base;
    if (!W || !p) return BSW_E_INVAL;
    bsw_default_params(p);                        /* matrix a=1,b=4,N=-1 is hard-wired (sw_extend.v:1915-1940) */
    p->o_del = (int)(W[0] & 0xff); p->e_del = (int)((W[0] >> 8) & 0xff);
    p->o_ins = (int)((W[0] >> 16) & 0xff); p->e_ins = (int)((W[0] >> 24) & 0xff);
    p->pen_clip5 = (int)(W[1] & 0xff); p->pen_clip3 = (int)((W[1] >> 8) & 0xff);
    p->w = (int)((W[1] >> 16) & 0xff);
    p->zdrop = 0;                                 /* the RTL has no zdrop port (Q3) */
    p->max_band_try = 2;
    n = W[2];
    if (n > BSW_REFBATCH_MAX_TASKS || 8 + 8 * (size_t)n > BSW_REFBATCH_IN_WORDS) return BSW_E_LIMIT;
    if (n == 0) return 0;
    if (!tasks || max_tasks < n || !seqbuf) return BSW_E_INVAL;
    base = (int64_t)(8 + 8 * n) - (int64_t)W[8 + 2];
    for (i = 0; i < n; ++i) {
        const uint32_t *H = &W[8 + 8 * i];
        bsw_task *t = &tasks[i];
        int len[4], s, nib = 0;
        int64_t pos = base + (int64_t)H[2];
        uint8_t *dst[4];
        memset(t, 0, sizeof(*t));
        t->lqlen = (int)(H[0] & 0xff); t->ltlen = (int)((H[0] >> 16) & 0x7ff);
        t->rqlen = (int)(H[1] & 0xff); t->rtlen = (int)((H[1] >> 16) & 0x7ff);
        t->qbeg = (int)(H[3] >> 16);
        t->init_score = (int)(int16_t)(H[3] & 0xffff);
        t->h0 = (int)(H[4] & 0xff);
        t->tag = H[7];
        /* H5/H6 {max_del[31:16], max_ins[15:0]}: the band limits the RTL applies (proc_element.v:925,933;
         * sw_extend.v:1881,1890).  A non-positive limit (never written by bwa: both are >= 1) decodes as 1. */
        {
            int mi = (int)(int16_t)(H[5] & 0xffff), md = (int)(int16_t)(H[5] >> 16);
            t->wlim_l = mi < md ? mi : md;
            if (t->wlim_l < 1) t->wlim_l = 1;
            mi = (int)(int16_t)(H[6] & 0xffff); md = (int)(int16_t)(H[6] >> 16);
            t->wlim_r = mi < md ? mi : md;
            if (t->wlim_r < 1) t->wlim_r = 1;
        }
        len[0] = t->lqlen; len[1] = t->rqlen; len[2] = t->ltlen; len[3] = t->rtlen;
        if (pos < 8 + 8 * (int64_t)n ||
            pos + (len[0] + len[1] + len[2] + len[3] + 7) / 8 > BSW_REFBATCH_IN_WORDS) return BSW_E_INVAL;
        if (used + (size_t)(len[0] + len[1] + len[2] + len[3]) > seqbuf_len) return BSW_E_NOMEM;
        for (s = 0; s < 4; ++s) { dst[s] = seqbuf + used; used += (size_t)len[s]; }
        t->lquery = dst[0]; t->rquery = dst[1]; t->ltarget = dst[2]; t->rtarget = dst[3];
        for (s = 0; s < 4; ++s) {
            int j;
            for (j = 0; j < len[s]; ++j, ++nib)
                dst[s][j] = (uint8_t)((W[pos + (nib >> 3)] >> (28 - 4 * (nib & 7))) & 0xf);
        }
    }
    return (int)n;
}

int bsw_refbatch_encode_results(const bsw_result *res, size_t n, uint32_t *W)
{
    size_t i;
    if (!W || (!res && n)) return BSW_E_INVAL;
    if (n > BSW_REFBATCH_MAX_TASKS) return BSW_E_LIMIT;
    for (i = 0; i < n; ++i) {
        const bsw_result *r = &res[i];
        uint32_t *R = &W[5 * i];
        R[0] = r->tag;
        R[1] = ((uint32_t)r->qe << 16) | ((uint32_t)r->qb & 0xffffu);
        R[2] = ((uint32_t)r->re << 16) | ((uint32_t)r->rb & 0xffffu);
        R[3] = ((uint32_t)r->truesc << 16) | ((uint32_t)r->score & 0xffffu);
        R[4] = (uint32_t)r->w;
    }
    return (int)n;
}

int bsw_refbatch_decode_results(const uint32_t *W, size_t n, bsw_result *res)
{
    size_t i;
    if (!W || (!res && n)) return BSW_E_INVAL;
    if (n > BSW_REFBATCH_MAX_TASKS) return BSW_E_LIMIT;
    for (i = 0; i < n; ++i) {
        const uint32_t *R = &W[5 * i];
        bsw_result *r = &res[i];
        memset(r, 0, sizeof(*r));
        r->tag = R[0];
        r->qb = (int16_t)(R[1] & 0xffff); r->qe = (int16_t)(R[1] >> 16);
        r->rb = (int16_t)(R[2] & 0xffff); r->re = (int16_t)(R[2] >> 16);
        r->score = (int16_t)(R[3] & 0xffff); r->truesc = (int16_t)(R[3] >> 16);
        r->w = (int)R[4];
    }
    return (int)n;
}
