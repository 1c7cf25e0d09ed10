/*
 * bsw_synth.c — synthetic seed-extension workload generator (host, plain C).
 *
 * The reference is driven by `bwa mem` against a genome (README.md:29-36); neither
 * bwa nor a genome exists in this image, so extension tasks are synthesised with
 * the shapes mem_chain2aln would hand to ksw_extend2:
 *   - a read of read_len bases, a seed of seed_len bases at qbeg,
 *   - left flank  : query[0..qbeg) reversed, reference left of the seed reversed,
 *   - right flank : query[qbeg+seed_len..), reference right of the seed,
 *   - tlen = qlen + cal_max_gap(qlen), cal_max_gap = min(max((qlen*a-o)/e+1,1), 2w).
 * Both flanks are produced "outward from the seed", which for i.i.d. sequence is
 * distributionally identical to reversing the left one.
 * PRNG: splitmix64 keyed by (seed, task index) so output is order-independent.
 */
#include "../../include/bwa_sw_mi355.h"

#include <string.h>

static inline uint64_t sm64(uint64_t *s)
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline double u01(uint64_t *s) { return (double)(sm64(s) >> 11) * (1.0 / 9007199254740992.0); }
static inline int urange(uint64_t *s, int lo, int hi) /* inclusive */
{
    return lo + (int)(sm64(s) % (uint64_t)(hi - lo + 1));
}

static int cal_max_gap(const bsw_synth_spec *sp, int qlen)
{
    int l = (int)((double)(qlen * sp->a - sp->o) / sp->e + 1.);
    if (l < 1) l = 1;
    return l < (sp->w << 1) ? l : (sp->w << 1);
}

size_t bsw_synth_arena_bound(const bsw_synth_spec *sp, size_t n)
{
    size_t per = (size_t)(4 * sp->read_len + 4 * sp->w + 32);
    return per * n + 64;
}

/* emit tlen reference bases and a qlen-base read derived from them */
static void make_flank(const bsw_synth_spec *sp, uint64_t *rs, int junk,
                       uint8_t *q, int qlen, uint8_t *t, int tlen)
{
    int i, rp = 0;
    for (i = 0; i < tlen; ++i) t[i] = (uint8_t)(sm64(rs) & 3);
    if (junk) {
        for (i = 0; i < qlen; ++i) q[i] = (uint8_t)(sm64(rs) & 3);
    } else {
        for (i = 0; i < qlen; ) {
            double r = u01(rs);
            if (r < sp->indel_rate * 0.5) {                 /* insertion in the read */
                q[i++] = (uint8_t)(sm64(rs) & 3);
            } else if (r < sp->indel_rate) {                /* deletion: skip reference bases */
                int len = 1; while (u01(rs) < 1.0 / 3.0 && len < 8) ++len;
                rp += len;
            } else {
                uint8_t b = rp < tlen ? t[rp] : (uint8_t)(sm64(rs) & 3);
                ++rp;
                if (u01(rs) < sp->sub_rate) b = (uint8_t)((b + 1 + (sm64(rs) % 3)) & 3);
                q[i++] = b;
            }
        }
    }
    if (sp->n_rate > 0) {
        for (i = 0; i < qlen; ++i) if (u01(rs) < sp->n_rate) q[i] = 4;
        for (i = 0; i < tlen; ++i) if (u01(rs) < sp->n_rate) t[i] = 4;
    }
}

int64_t bsw_synth_generate(const bsw_synth_spec *sp, size_t n, bsw_task *tasks, uint8_t *arena, size_t arena_len)
{
    size_t k, off = 0;
    if (!sp || !tasks || !arena || sp->read_len < 2 || sp->seed_len_min < 1 ||
        sp->seed_len_max < sp->seed_len_min || sp->seed_len_max >= sp->read_len || sp->e < 1 || sp->a < 1)
        return BSW_E_INVAL;
    for (k = 0; k < n; ++k) {
        uint64_t rs = sp->seed * 0xD1342543DE82EF95ull + (uint64_t)k * 0x2545F4914F6CDD1Dull + 1;
        bsw_task *t = &tasks[k];
        int sl = urange(&rs, sp->seed_len_min, sp->seed_len_max);
        int qbeg = sp->seed_at_start ? 0 : urange(&rs, 0, sp->read_len - sl);
        int junk = u01(&rs) < sp->junk_frac;
        int lq = qbeg, rq = sp->read_len - qbeg - sl;
        int lt = lq ? lq + cal_max_gap(sp, lq) : 0;
        int rt = rq ? rq + cal_max_gap(sp, rq) : 0;
        if (off + (size_t)(lq + lt + rq + rt) > arena_len) return BSW_E_NOMEM;
        memset(t, 0, sizeof(*t));
        t->lqlen = lq; t->ltlen = lt; t->rqlen = rq; t->rtlen = rt;
        t->lquery = arena + off;  off += (size_t)lq;
        t->ltarget = arena + off; off += (size_t)lt;
        t->rquery = arena + off;  off += (size_t)rq;
        t->rtarget = arena + off; off += (size_t)rt;
        if (lq) make_flank(sp, &rs, junk, (uint8_t *)t->lquery, lq, (uint8_t *)t->ltarget, lt);
        if (rq) make_flank(sp, &rs, junk, (uint8_t *)t->rquery, rq, (uint8_t *)t->rtarget, rt);
        t->h0 = sl * sp->a;
        t->init_score = -1;
        t->qbeg = qbeg;
        t->tag = (uint32_t)k;
    }
    return (int64_t)off;
}

/* ---- synthetic genome + reads for the device-resident-reference path (F3) -----------------------------------
 * pac: (l_pac+3)/4 bytes, bwa's .pac layout (4 bases per byte, first base in the top two bits), i.i.d. bases.
 * Read k (forward strand): one exact seed of seed_len bases copied from the genome at rbeg, flanks derived from
 * the genome outward from the seed with the spec's substitution / indel / N rates (junk_frac: unrelated flanks);
 * rmax = the chain window bwa's mem_chain2aln would use for this single-seed chain (bsw_chain_window). */
static inline uint8_t pac_at(const uint8_t *pac, int64_t l) { return (uint8_t)((pac[l >> 2] >> ((~l & 3) << 1)) & 3); }

static void derive_flank(const bsw_synth_spec *sp, uint64_t *rs, int junk, uint8_t *q, int qlen, const uint8_t *pac, int64_t t0, int dir, int tmax)
{
    int i, rp = 0;
    if (junk) {
        for (i = 0; i < qlen; ++i) q[i] = (uint8_t)(sm64(rs) & 3);
    } else {
        for (i = 0; i < qlen; ) {
            double r = u01(rs);
            if (r < sp->indel_rate * 0.5) {
                q[i++] = (uint8_t)(sm64(rs) & 3);
            } else if (r < sp->indel_rate) {
                int len = 1; while (u01(rs) < 1.0 / 3.0 && len < 8) ++len;
                rp += len;
            } else {
                uint8_t b = rp < tmax ? pac_at(pac, t0 + (int64_t)dir * rp) : (uint8_t)(sm64(rs) & 3);
                ++rp;
                if (u01(rs) < sp->sub_rate) b = (uint8_t)((b + 1 + (sm64(rs) % 3)) & 3);
                q[i++] = b;
            }
        }
    }
    if (sp->n_rate > 0)
        for (i = 0; i < qlen; ++i) if (u01(rs) < sp->n_rate) q[i] = 4;
}

int64_t bsw_synth_ref_generate(const bsw_synth_spec *sp, const bsw_params *p, int64_t l_pac, uint8_t *pac, size_t n,
                               bsw_ref_task *rt, uint8_t *arena, size_t arena_len)
{
    size_t k;
    int64_t i;
    uint64_t gs;
    const int64_t margin = 2 * (int64_t)(sp ? sp->read_len : 0) + 4 * (int64_t)(sp ? sp->w : 0) + 64;
    if (!sp || !p || !pac || (!rt && n) || (!arena && n) || sp->read_len < 2 || sp->seed_len_min < 1 ||
        sp->seed_len_max < sp->seed_len_min || sp->seed_len_max >= sp->read_len || l_pac < 4 * margin)
        return BSW_E_INVAL;
    if ((size_t)sp->read_len * n > arena_len) return BSW_E_NOMEM;
    memset(pac, 0, (size_t)((l_pac + 3) >> 2));
    gs = sp->seed * 0x9E3779B97F4A7C15ull + 12345;
    for (i = 0; i < l_pac; i += 32) {                      /* 32 bases per draw */
        uint64_t z = sm64(&gs);
        int j;
        for (j = 0; j < 32 && i + j < l_pac; ++j) pac[(i + j) >> 2] |= (uint8_t)(((z >> (2 * j)) & 3) << ((~(i + j) & 3) << 1));
    }
    for (k = 0; k < n; ++k) {
        uint64_t rs = sp->seed * 0xD1342543DE82EF95ull + (uint64_t)k * 0x2545F4914F6CDD1Dull + 7;
        bsw_ref_task *t = &rt[k];
        uint8_t *read = arena + k * (size_t)sp->read_len;
        const int sl = urange(&rs, sp->seed_len_min, sp->seed_len_max);
        const int qbeg = sp->seed_at_start ? 0 : urange(&rs, 0, sp->read_len - sl);
        const int junk = u01(&rs) < sp->junk_frac;
        const int rq = sp->read_len - qbeg - sl;
        const int64_t rbeg = margin + (int64_t)(sm64(&rs) % (uint64_t)(l_pac - 2 * margin));
        int64_t rmax[2];
        int j;
        for (j = 0; j < sl; ++j) read[qbeg + j] = pac_at(pac, rbeg + j);
        if (qbeg) {                                        /* left flank: outward = backwards in the genome, then mirrored into the read */
            uint8_t tmp[1024];
            if (qbeg > 1024) return BSW_E_LIMIT;
            derive_flank(sp, &rs, junk, tmp, qbeg, pac, rbeg - 1, -1, (int)margin);
            for (j = 0; j < qbeg; ++j) read[qbeg - 1 - j] = tmp[j];
        }
        if (rq) derive_flank(sp, &rs, junk, read + qbeg + sl, rq, pac, rbeg + sl, 1, (int)margin);
        memset(t, 0, sizeof(*t));
        t->query = read; t->l_query = sp->read_len; t->init_score = -1;
        t->seed.rbeg = rbeg; t->seed.qbeg = qbeg; t->seed.len = sl;
        if (bsw_chain_window(p, &t->seed, 1, sp->read_len, l_pac, rmax) != BSW_OK) return BSW_E_INVAL;
        t->rmax0 = rmax[0]; t->rmax1 = rmax[1];
        t->tag = (uint32_t)k;
    }
    return (int64_t)((size_t)sp->read_len * n);
}
