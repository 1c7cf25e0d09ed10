/* Block-dispatch statistics of the lane kernels, from the CPU oracle's per-row [beg,end) ranges
 * (analysis tool, not product code): how many 8-column blocks a wave of S seeds runs dense / edge per row.
 * gcc -O2 -I../../include -o block_stats block_stats.c ../../bwa-mem-sw_amd/csrc/bsw_synth.c -lm */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "bwa_sw_mi355.h"

typedef struct { int h, e; } eh_t;
#define MAXROWS 1024
typedef struct { int nrows; short beg[MAXROWS], end[MAXROWS]; int qlen; long key; } trace_t;

/* variant H ksw_extend2, recording the clamped [beg,end) of every row it iterates */
static void extend_trace(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat,
                         int o_del, int e_del, int o_ins, int e_ins, int w, int end_bonus, int zdrop, int h0, trace_t *tr)
{
    eh_t *eh = calloc((size_t)qlen + 2, sizeof(eh_t));
    int oe_del = o_del + e_del, oe_ins = o_ins + e_ins, i, j, beg, end, max, max_i, max_j, max_ie, gscore, max_off, max_ins, max_del, k;
    eh[0].h = h0; if (qlen >= 1) eh[1].h = h0 > oe_ins ? h0 - oe_ins : 0;
    for (j = 2; j <= qlen && eh[j - 1].h > e_ins; ++j) eh[j].h = eh[j - 1].h - e_ins;
    for (i = 0, max = 0, k = 25; i < k; ++i) max = max > mat[i] ? max : mat[i];
    max_ins = (int)((double)(qlen * max + end_bonus - o_ins) / e_ins + 1.); if (max_ins < 1) max_ins = 1; if (w > max_ins) w = max_ins;
    max_del = (int)((double)(qlen * max + end_bonus - o_del) / e_del + 1.); if (max_del < 1) max_del = 1; if (w > max_del) w = max_del;
    max = h0; max_i = max_j = -1; max_ie = -1; gscore = -1; max_off = 0; beg = 0; end = qlen;
    tr->nrows = 0; tr->qlen = qlen;
    for (i = 0; i < tlen; ++i) {
        int f = 0, h1, mrow = 0, mj = -1;
        const int8_t *srow = &mat[target[i] * 5];
        if (beg < i - w) beg = i - w;
        if (end > i + w + 1) end = i + w + 1;
        if (end > qlen) end = qlen;
        if (beg == 0) { h1 = h0 - (o_del + e_del * (i + 1)); if (h1 < 0) h1 = 0; } else h1 = 0;
        if (tr->nrows < MAXROWS) { tr->beg[tr->nrows] = (short)beg; tr->end[tr->nrows] = (short)end; tr->nrows++; }
        for (j = beg; j < end; ++j) {
            eh_t *p = &eh[j]; int h = p->h, e = p->e, s = srow[query[j]], t;
            p->h = h1; h += s; h = h > e ? h : e; h = h > f ? h : f; h1 = h;
            mj = mrow > h ? mj : j; mrow = mrow > h ? mrow : h;
            t = h - oe_del; if (t < 0) t = 0; e -= e_del; if (e < t) e = t; p->e = e;
            t = h - oe_ins; if (t < 0) t = 0; f -= e_ins; if (f < t) f = t;
        }
        eh[end].h = h1; eh[end].e = 0;
        if (j == qlen) { max_ie = gscore > h1 ? max_ie : i; gscore = gscore > h1 ? gscore : h1; }
        if (mrow == 0) break;
        if (mrow > max) { int off = mj - i; if (off < 0) off = -off; max = mrow; max_i = i; max_j = mj; if (off > max_off) max_off = off; }
        else if (zdrop > 0) {
            if (i - max_i > mj - max_j) { if (max - mrow - ((i - max_i) - (mj - max_j)) * e_del > zdrop) break; }
            else { if (max - mrow - ((mj - max_j) - (i - max_i)) * e_ins > zdrop) break; }
        }
        for (j = beg; j < end && eh[j].h == 0 && eh[j].e == 0; ++j) {}
        beg = j;
        for (j = end; j >= beg && eh[j].h == 0 && eh[j].e == 0; --j) {}
        end = j + 2 < qlen ? j + 2 : qlen;
    }
    free(eh);
}

static int cmp_q(const void *a, const void *b) {
    const trace_t *x = a, *y = b;
    if (x->qlen != y->qlen) return y->qlen - x->qlen;
    return x->key < y->key ? -1 : x->key > y->key;
}

int main(int argc, char **argv)
{
    int n = argc > 1 ? atoi(argv[1]) : 8192, mode = argc > 2 ? atoi(argv[2]) : 0, S;
    bsw_params p; memset(&p, 0, sizeof(p));
    for (int i = 0; i < 5; ++i) for (int j = 0; j < 5; ++j) p.mat[i * 5 + j] = (i == 4 || j == 4) ? -1 : (i == j ? 1 : -4);
    p.w = 100;
    bsw_synth_spec sp; memset(&sp, 0, sizeof(sp));
    sp.seed = 1000; sp.read_len = 150; sp.seed_len_min = 19; sp.seed_len_max = 19; sp.seed_at_start = 1; sp.sub_rate = 0.01; sp.indel_rate = 0.001;
    sp.a = 1; sp.w = 100; sp.o = 6; sp.e = 1;
    if (mode == 1) { sp.seed_len_max = 60; sp.seed_at_start = 0; sp.junk_frac = 0.05; sp.n_rate = 0.0005; }
    if (mode == 2) { sp.read_len = 250; sp.seed_len_max = 40; sp.seed_at_start = 0; sp.sub_rate = 0.04; sp.indel_rate = 0.01; sp.junk_frac = 0.05; sp.w = 500; p.w = 500; }
    uint8_t *arena = malloc(bsw_synth_arena_bound(&sp, n));
    bsw_task *tasks = calloc(n, sizeof(bsw_task));
    bsw_synth_generate(&sp, n, tasks, arena, bsw_synth_arena_bound(&sp, n));
    trace_t *tr = calloc(n, sizeof(trace_t));
    int nt = 0;
    for (int k = 0; k < n; ++k) {           /* right sides only (h0 = seed score: left score unknown here, close enough) */
        if (!tasks[k].rqlen) continue;
        extend_trace(tasks[k].rqlen, tasks[k].rquery, tasks[k].rtlen, tasks[k].rtarget, p.mat, 6, 1, 6, 1, p.w, 5, 100, tasks[k].h0, &tr[nt]);
        {
            trace_t *t = &tr[nt]; long key = 0;
            int kmode = argc > 3 ? atoi(argv[3]) : 0;
            if (kmode == 1) { for (int i = 0; i < t->nrows; ++i) key += t->end[i] - t->beg[i]; }          /* total cells (oracle knowledge) */
            if (kmode == 2) { int L = tasks[k].rqlen < tasks[k].rtlen ? tasks[k].rqlen : tasks[k].rtlen; for (int i = 0; i < L; ++i) key += tasks[k].rquery[i] != tasks[k].rtarget[i]; }   /* diagonal mismatches */
            if (kmode == 3) { key = t->nrows; }
            if (kmode == 4) { int L = tasks[k].rqlen < tasks[k].rtlen ? tasks[k].rqlen : tasks[k].rtlen; int i; for (i = 0; i < L && tasks[k].rquery[i] == tasks[k].rtarget[i]; ++i) {} key = i; }  /* first mismatch */
            if (kmode == 5) { int L = tasks[k].rqlen < tasks[k].rtlen ? tasks[k].rqlen : tasks[k].rtlen; int sc = tasks[k].h0, mn = 1000; for (int i = 0; i < L; ++i) { sc += tasks[k].rquery[i] == tasks[k].rtarget[i] ? 1 : -4; if (sc < mn) mn = sc; } key = mn; }
            if (kmode == 6) key = -tasks[k].h0;
            if (kmode == 7) key = -(tasks[k].h0 / 4);
            t->key = key;
        }
        ++nt;
    }
    qsort(tr, nt, sizeof(trace_t), cmp_q);
    for (S = 64; S <= 128; S += 64) {
        double cells = 0, dense = 0, edge = 0, rows = 0, lanerows = 0, eleft = 0, eru = 0, egen = 0, bandrows = 0;
        for (int w0 = 0; w0 + S <= nt; w0 += S) {
            int maxrows = 0;
            for (int l = 0; l < S; ++l) if (tr[w0 + l].nrows > maxrows) maxrows = tr[w0 + l].nrows;
            for (int i = 0; i < maxrows; ++i) {
                int jlo = 1 << 20, jhi = -1, jbm = -1, jem = 1 << 20, act = 0;
                for (int l = 0; l < S; ++l) {
                    trace_t *t = &tr[w0 + l];
                    if (i >= t->nrows) continue;
                    ++act;
                    int b = t->beg[i], e = t->end[i];
                    cells += e > b ? e - b : 0;
                    if (b < jlo) jlo = b; if (e > jhi) jhi = e; if (b > jbm) jbm = b; if (e < jem) jem = e;
                }
                rows += 1; lanerows += act;
                for (int j0 = 0; j0 < 232; j0 += 8) {
                    if (j0 + 8 <= jlo || j0 > jhi) continue;
                    if (j0 >= jbm && j0 + 8 <= jem) dense += 1;
                    else {
                        edge += 1;
                        if (j0 + 8 <= jem) eleft += 1;                       /* only ragged on the left */
                        else if (jem == jhi && j0 >= jbm) eru += 1;          /* right edge, same `end` in every lane */
                        else egen += 1;
                    }
                }
            }
        }
        printf("   edge split: left-only %.2f, right-uniform %.2f, general %.2f per row\n", eleft / rows, eru / rows, egen / rows);
        printf("seeds/wave %3d: rows %.0f, active lanes/row %.1f, blocks/row: dense %.2f edge %.2f; lane-cell slots %.3g vs cells %.3g -> lane efficiency %.3f\n",
               S, rows, lanerows / rows, dense / rows, edge / rows, (dense + edge) * 8 * S, cells, cells / ((dense + edge) * 8 * S));
    }
    return 0;
}
