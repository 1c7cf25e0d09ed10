/* Sanitizer run of the plain-C host code (reference wire-format codec, mem_chain2aln glue, generator) and of
 * the CPU oracle: compiled with -fsanitize=address,undefined by tests/test_sanitizers_cpu.py.  GPU sanitizers are
 * not available on the pool, so the host-side C code is what gets this treatment. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "bwa_sw_mi355.h"
#include "../oracle/ksw_extend_ref.h"

void bsw_default_params(bsw_params *p)           /* the real one lives in the HIP translation unit */
{
    memset(p, 0, sizeof(*p));
    for (int i = 0; i < 5; ++i) for (int j = 0; j < 5; ++j) p->mat[i * 5 + j] = (i == 4 || j == 4) ? -1 : (i == j ? 1 : -4);
    p->o_del = p->o_ins = 6; p->e_del = p->e_ins = 1; p->w = 100; p->pen_clip5 = p->pen_clip3 = 5; p->zdrop = 100; p->max_band_try = 2;
}

int main(void)
{
    bsw_params p; bsw_default_params(&p);
    bsw_synth_spec sp; memset(&sp, 0, sizeof(sp));
    sp.seed = 9; sp.read_len = 150; sp.seed_len_min = 19; sp.seed_len_max = 60; sp.sub_rate = 0.02; sp.indel_rate = 0.01;
    sp.n_rate = 0.01; sp.junk_frac = 0.1; sp.a = 1; sp.w = 100; sp.o = 6; sp.e = 1;
    const size_t n = 1500;
    size_t cap = bsw_synth_arena_bound(&sp, n);
    uint8_t *arena = malloc(cap);
    bsw_task *tasks = malloc(n * sizeof(*tasks));
    bsw_result *res = malloc(n * sizeof(*res));
    if (bsw_synth_generate(&sp, n, tasks, arena, cap) < 0) return 1;
    bsw_pair_batch_ref(&p, tasks, n, res, 3);                               /* oracle, 3 threads */
    for (int variant = 0; variant < 2; ++variant) { p.variant = variant; bsw_pair_batch_ref(&p, tasks, 200, res, 1); }
    /* wire format round trip */
    uint32_t *W = malloc(BSW_REFBATCH_IN_WORDS * 4), *R = calloc(BSW_REFBATCH_OUT_WORDS, 4);
    int k = bsw_refbatch_encode(&p, tasks, n, W);
    if (k <= 0) return 2;
    bsw_params p2; bsw_task *t2 = malloc(BSW_REFBATCH_MAX_TASKS * sizeof(*t2));
    uint8_t *sb = malloc((size_t)BSW_REFBATCH_IN_WORDS * 8 + 64);
    if (bsw_refbatch_decode(W, &p2, t2, BSW_REFBATCH_MAX_TASKS, sb, (size_t)BSW_REFBATCH_IN_WORDS * 8 + 64) != k) return 3;
    for (int i = 0; i < k; ++i)
        if (t2[i].rqlen != tasks[i].rqlen || memcmp(t2[i].rquery, tasks[i].rquery, (size_t)t2[i].rqlen)) return 4;
    bsw_pair_batch_ref(&p2, t2, (size_t)k, res, 1);
    if (bsw_refbatch_encode_results(res, (size_t)k, R) != k || bsw_refbatch_decode_results(R, (size_t)k, res) != k) return 5;
    W[2] = 100000; if (bsw_refbatch_decode(W, &p2, t2, BSW_REFBATCH_MAX_TASKS, sb, 10) >= 0) return 6;      /* malformed */
    /* glue on a tiny genome, both strands */
    enum { LP = 4000 };
    uint8_t *g = malloc(LP), *pac = calloc(LP / 4 + 1, 1), *rseq = malloc(2 * LP), scratch[4096];
    for (int i = 0; i < LP; ++i) { g[i] = (uint8_t)((i * 7 + i / 3) & 3); pac[i >> 2] |= (uint8_t)(g[i] << ((~i & 3) << 1)); }
    for (int it = 0; it < 200; ++it) {
        bsw_seed s = { (it & 1 ? LP : 0) + 200 + it * 13 % 3000, it % 100, 19 + it % 30 };
        int64_t rmax[2];
        uint8_t read[150];
        int64_t got = bsw_pac_get_seq(LP, pac, s.rbeg - s.qbeg, s.rbeg - s.qbeg + 150, read);
        if (got != 150) continue;
        if (bsw_chain_window(&p, &s, 1, 150, LP, rmax)) return 7;
        int64_t rl = bsw_pac_get_seq(LP, pac, rmax[0], rmax[1], rseq);
        if (rl != rmax[1] - rmax[0]) return 8;
        bsw_task t; bsw_result r; bsw_alnreg a;
        if (bsw_seed_to_task(&p, &s, 150, read, rmax[0], rmax[1], rseq, scratch, sizeof(scratch), (uint32_t)it, &t)) return 9;
        p.variant = 0;
        bsw_pair_ref(&p, &t, &r);
        bsw_result_to_alnreg(&s, &r, &a);
        if (a.qb != 0 || a.qe != 150 || a.rb != s.rbeg - s.qbeg || a.score != 150) return 10;
    }
    /* synthetic genome + reads: every read extends inside its own chain window */
    {
        enum { LG = 20000, NR = 300 };
        uint8_t *gp = malloc(LG / 4 + 1), *ar = malloc((size_t)NR * 150), *win = malloc(1024), sc2[4096];
        bsw_ref_task *rt = malloc(NR * sizeof(*rt));
        p.variant = 0;
        if (bsw_synth_ref_generate(&sp, &p, LG, gp, NR, rt, ar, (size_t)NR * 150) < 0) return 11;
        if (bsw_synth_ref_generate(&sp, &p, LG, gp, NR, rt, ar, 10) >= 0) return 12;                        /* arena too small */
        for (int i = 0; i < NR; ++i) {
            int64_t rl = bsw_pac_get_seq(LG, gp, rt[i].rmax0, rt[i].rmax1, win);
            if (rl != rt[i].rmax1 - rt[i].rmax0) return 13;
            bsw_task t; bsw_result r;
            if (bsw_seed_to_task(&p, &rt[i].seed, rt[i].l_query, rt[i].query, rt[i].rmax0, rt[i].rmax1, win, sc2, sizeof(sc2), (uint32_t)i, &t)) return 14;
            bsw_pair_ref(&p, &t, &r);
            if (r.score < rt[i].seed.len * p.mat[0] - p.pen_clip5 - p.pen_clip3) return 15;
        }
        free(gp); free(ar); free(win); free(rt);
    }
    free(arena); free(tasks); free(res); free(W); free(R); free(t2); free(sb); free(g); free(pac); free(rseq);
    puts("asan_host ok");
    return 0;
}
