/*
 * bsw_host_example.c — minimal C host over the C ABI (what INTEGRATION.md describes).
 * Mirrors the reference host's command line where it applies (README.md:29-36 of the reference:
 * `bwa --target=ASE|Direct mem -t N -b BATCH ...`) with bwa's scoring flags:
 *   bsw_host_example [-A a] [-B b] [-O o] [-E e] [-L clip] [-w band] [-d zdrop] [-b batch_seeds]
 *                    [-n seeds] [-l read_len] [-g gpu] [--variant=H|M]
 * Generates synthetic seeds, streams them through bsw_submit/bsw_wait, prints seeds/s and GCUPS.
 * Build: gcc -O2 -Iinclude tools/bsw_host_example.c -Lbwa-mem-sw_amd -lbwasw_mi355 -Wl,-rpath,$PWD/bwa-mem-sw_amd -o bsw_host_example
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "bwa_sw_mi355.h"

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }

int main(int argc, char **argv)
{
    int a = 1, b = 4, o = 6, e = 1, clip = 5, w = 100, zdrop = 100, gpu = 0, read_len = 150, variant = BSW_VARIANT_H;
    size_t n = 200000, batch = 65536, i;
    for (int k = 1; k < argc; ++k) {
        const char *f = argv[k], *v = k + 1 < argc ? argv[k + 1] : "";
        if (!strcmp(f, "-A")) a = atoi(v), ++k; else if (!strcmp(f, "-B")) b = atoi(v), ++k;
        else if (!strcmp(f, "-O")) o = atoi(v), ++k; else if (!strcmp(f, "-E")) e = atoi(v), ++k;
        else if (!strcmp(f, "-L")) clip = atoi(v), ++k; else if (!strcmp(f, "-w")) w = atoi(v), ++k;
        else if (!strcmp(f, "-d")) zdrop = atoi(v), ++k; else if (!strcmp(f, "-b")) batch = (size_t)atol(v), ++k;
        else if (!strcmp(f, "-n")) n = (size_t)atol(v), ++k; else if (!strcmp(f, "-l")) read_len = atoi(v), ++k;
        else if (!strcmp(f, "-g")) gpu = atoi(v), ++k;
        else if (!strcmp(f, "--variant=M")) variant = BSW_VARIANT_M; else if (!strcmp(f, "--variant=H")) variant = BSW_VARIANT_H;
        else { fprintf(stderr, "unknown flag %s\n", f); return 2; }
    }
    bsw_params p; bsw_default_params(&p);
    for (int r = 0; r < 5; ++r) for (int c = 0; c < 5; ++c) p.mat[r * 5 + c] = (r == 4 || c == 4) ? -1 : (r == c ? a : -b);
    p.o_del = p.o_ins = o; p.e_del = p.e_ins = e; p.pen_clip5 = p.pen_clip3 = clip; p.w = w; p.zdrop = zdrop; p.variant = variant;

    bsw_synth_spec sp; memset(&sp, 0, sizeof(sp));
    sp.seed = 1; sp.read_len = read_len; sp.seed_len_min = 19; sp.seed_len_max = 60; sp.seed_at_start = 0;
    sp.sub_rate = 0.01; sp.indel_rate = 0.001; sp.junk_frac = 0.05; sp.a = a; sp.w = w; sp.o = o; sp.e = e;
    size_t cap = bsw_synth_arena_bound(&sp, n);
    uint8_t *arena = malloc(cap); bsw_task *tasks = malloc(n * sizeof(*tasks)); bsw_result *res = malloc(n * sizeof(*res));
    if (!arena || !tasks || !res || bsw_synth_generate(&sp, n, tasks, arena, cap) < 0) { fprintf(stderr, "generator failed\n"); return 1; }

    bsw_config cfg; bsw_default_config(&cfg); cfg.device = gpu; cfg.chunk_tasks = batch;
    bsw_ctx *ctx = NULL;
    int rc = bsw_create(&cfg, &ctx);
    if (rc != BSW_OK) { fprintf(stderr, "bsw_create failed (%d): no gfx950 GPU, and this library has no CPU path\n", rc); return 1; }
    double t0 = now();
    rc = bsw_submit(ctx, &p, tasks, n, res);
    if (rc == BSW_OK) rc = bsw_wait(ctx);
    double dt = now() - t0;
    if (rc != BSW_OK) { fprintf(stderr, "GPU path failed (%d): %s\n", rc, bsw_last_error(ctx)); return 1; }
    unsigned long long cells = 0;
    for (i = 0; i < n; ++i) cells += res[i].left.cells + res[i].right.cells;
    printf("{\"seeds\": %zu, \"seconds\": %.4f, \"seeds_per_s\": %.1f, \"gcups_incl_pack_and_pcie\": %.2f}\n",
           n, dt, n / dt, cells / dt / 1e9);
    bsw_destroy(ctx); free(arena); free(tasks); free(res);
    return 0;
}
