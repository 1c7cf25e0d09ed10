/*
 * bsw_host_example.c — "bsw-bench": a plain-C host over the C ABI (what INTEGRATION.md describes).
 *
 * Mirrors the reference host's command line where it applies (reference README.md:29-36:
 * `bwa --target=ASE|Direct mem -t N -b BATCH ...`) with bwa mem's scoring flags:
 *
 *   bsw-bench [--target=hip] [--gpus G] [-t pack_threads] [-b batch_seeds] [-n seeds] [-l read_len]
 *             [-A a] [-B b] [-O o] [-E e] [-L clip] [-w band] [-d zdrop] [--variant=H|M]
 *             [--dump FILE | --load FILE]
 *
 * --target=cpu is refused: the library has no CPU path (the CPU oracle lives under oracle/ and is test-only).
 * --gpus G shards the seed pool by read over G contexts, one host thread per GPU, task k -> GPU (k / batch) mod G,
 * exactly like the reference's round-robin over its 4 PE arrays (batch_manager.v:343-348) — no inter-GPU traffic.
 * --dump / --load write / read a self-contained task batch (params + seeds + sequences) for reproducible runs.
 *
 * Build: gcc -O2 -Iinclude tools/bsw_host_example.c -Lbwa-mem-sw_amd -lbwasw_mi355 -lpthread \
 *            -Wl,-rpath,$PWD/bwa-mem-sw_amd -o bsw-bench
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "bwa_sw_mi355.h"

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }

typedef struct {
    int gpu, pack_threads, rc;
    size_t batch, n;
    const bsw_params *p;
    bsw_task *tasks;       /* this GPU's shard, gathered */
    bsw_result *res;
    char err[256];
} shard_t;

static void *shard_main(void *arg)
{
    shard_t *s = (shard_t *)arg;
    bsw_config cfg; bsw_default_config(&cfg);
    cfg.device = s->gpu; cfg.chunk_tasks = s->batch; cfg.pack_threads = s->pack_threads; cfg.streams = 3;
    bsw_ctx *ctx = NULL;
    s->rc = bsw_create(&cfg, &ctx);
    if (s->rc != BSW_OK) { snprintf(s->err, sizeof(s->err), "bsw_create on GPU %d failed (%d): no CPU path exists", s->gpu, s->rc); return NULL; }
    s->rc = bsw_submit(ctx, s->p, s->tasks, s->n, s->res);          /* = ring CSR_REQ_PEARRAY                */
    if (s->rc == BSW_OK) s->rc = bsw_wait(ctx);                     /* = poll the DSM busy bit              */
    if (s->rc != BSW_OK) snprintf(s->err, sizeof(s->err), "GPU %d: %s", s->gpu, bsw_last_error(ctx));
    bsw_destroy(ctx);
    return NULL;
}

/* dump format: magic, bsw_params, n, per seed {4 lengths, h0, init_score, qbeg, tag}, then all sequences lq|lt|rq|rt */
static int dump_batch(const char *path, const bsw_params *p, const bsw_task *t, size_t n)
{
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    const uint64_t magic = 0x3148435441425753ull, nn = n;            /* "SWBATCH1" */
    fwrite(&magic, 8, 1, f); fwrite(p, sizeof(*p), 1, f); fwrite(&nn, 8, 1, f);
    for (size_t i = 0; i < n; ++i) {
        int32_t h[8] = {t[i].lqlen, t[i].ltlen, t[i].rqlen, t[i].rtlen, t[i].h0, t[i].init_score, t[i].qbeg, (int32_t)t[i].tag};
        fwrite(h, sizeof(h), 1, f);
    }
    for (size_t i = 0; i < n; ++i) {
        if (t[i].lqlen) { fwrite(t[i].lquery, 1, (size_t)t[i].lqlen, f); fwrite(t[i].ltarget, 1, (size_t)t[i].ltlen, f); }
        if (t[i].rqlen) { fwrite(t[i].rquery, 1, (size_t)t[i].rqlen, f); fwrite(t[i].rtarget, 1, (size_t)t[i].rtlen, f); }
    }
    return fclose(f);
}

static int load_batch(const char *path, bsw_params *p, bsw_task **tasks, uint8_t **arena, size_t *n)
{
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    uint64_t magic = 0, nn = 0;
    if (fread(&magic, 8, 1, f) != 1 || magic != 0x3148435441425753ull || fread(p, sizeof(*p), 1, f) != 1 || fread(&nn, 8, 1, f) != 1) { fclose(f); return -2; }
    bsw_task *t = calloc(nn ? nn : 1, sizeof(*t));
    size_t total = 0;
    for (size_t i = 0; i < nn; ++i) {
        int32_t h[8];
        if (fread(h, sizeof(h), 1, f) != 1) { fclose(f); return -3; }
        t[i].lqlen = h[0]; t[i].ltlen = h[1]; t[i].rqlen = h[2]; t[i].rtlen = h[3];
        t[i].h0 = h[4]; t[i].init_score = h[5]; t[i].qbeg = h[6]; t[i].tag = (uint32_t)h[7];
        total += (size_t)(h[0] ? h[0] + h[1] : 0) + (size_t)(h[2] ? h[2] + h[3] : 0);
    }
    uint8_t *a = malloc(total + 1);
    if (fread(a, 1, total, f) != total) { fclose(f); return -4; }
    size_t off = 0;
    for (size_t i = 0; i < nn; ++i) {
        if (t[i].lqlen) { t[i].lquery = a + off; off += (size_t)t[i].lqlen; t[i].ltarget = a + off; off += (size_t)t[i].ltlen; }
        if (t[i].rqlen) { t[i].rquery = a + off; off += (size_t)t[i].rqlen; t[i].rtarget = a + off; off += (size_t)t[i].rtlen; }
    }
    fclose(f);
    *tasks = t; *arena = a; *n = nn;
    return 0;
}

int main(int argc, char **argv)
{
    int a = 1, b = 4, o = 6, e = 1, clip = 5, w = 100, zdrop = 100, gpus = 1, read_len = 150, variant = BSW_VARIANT_H, threads = 8;
    size_t n = 200000, batch = 65536, i;
    const char *dump = NULL, *load = NULL;
    for (int k = 1; k < argc; ++k) {
        const char *f = argv[k], *v = k + 1 < argc ? argv[k + 1] : "";
        if (!strcmp(f, "-A")) a = atoi(v), ++k; else if (!strcmp(f, "-B")) b = atoi(v), ++k;
        else if (!strcmp(f, "-O")) o = atoi(v), ++k; else if (!strcmp(f, "-E")) e = atoi(v), ++k;
        else if (!strcmp(f, "-L")) clip = atoi(v), ++k; else if (!strcmp(f, "-w")) w = atoi(v), ++k;
        else if (!strcmp(f, "-d")) zdrop = atoi(v), ++k; else if (!strcmp(f, "-b")) batch = (size_t)atol(v), ++k;
        else if (!strcmp(f, "-n")) n = (size_t)atol(v), ++k; else if (!strcmp(f, "-l")) read_len = atoi(v), ++k;
        else if (!strcmp(f, "-t")) threads = atoi(v), ++k; else if (!strcmp(f, "--gpus")) gpus = atoi(v), ++k;
        else if (!strcmp(f, "--dump")) dump = v, ++k; else if (!strcmp(f, "--load")) load = v, ++k;
        else if (!strcmp(f, "--variant=M")) variant = BSW_VARIANT_M; else if (!strcmp(f, "--variant=H")) variant = BSW_VARIANT_H;
        else if (!strcmp(f, "--target=hip")) {}
        else if (!strncmp(f, "--target=", 9)) { fprintf(stderr, "%s: only --target=hip exists; this library has no CPU path\n", f); return 2; }
        else { fprintf(stderr, "unknown flag %s\n", f); return 2; }
    }
    if (gpus < 1 || batch < 1) return 2;
    bsw_params p; bsw_default_params(&p);
    bsw_task *tasks = NULL; uint8_t *arena = NULL;
    if (load) {
        int rc = load_batch(load, &p, &tasks, &arena, &n);
        if (rc) { fprintf(stderr, "cannot load %s (%d)\n", load, rc); return 1; }
    } else {
        for (int r = 0; r < 5; ++r) for (int c = 0; c < 5; ++c) p.mat[r * 5 + c] = (r == 4 || c == 4) ? -1 : (r == c ? a : -b);
        p.o_del = p.o_ins = o; p.e_del = p.e_ins = e; p.pen_clip5 = p.pen_clip3 = clip; p.w = w; p.zdrop = zdrop; p.variant = variant;
        bsw_synth_spec sp; memset(&sp, 0, sizeof(sp));
        sp.seed = 1; sp.read_len = read_len; sp.seed_len_min = 19; sp.seed_len_max = 60; sp.seed_at_start = 0;
        sp.sub_rate = 0.01; sp.indel_rate = 0.001; sp.junk_frac = 0.05; sp.a = a; sp.w = w; sp.o = o; sp.e = e;
        size_t cap = bsw_synth_arena_bound(&sp, n);
        arena = malloc(cap); tasks = malloc((n ? n : 1) * sizeof(*tasks));
        if (!arena || !tasks || bsw_synth_generate(&sp, n, tasks, arena, cap) < 0) { fprintf(stderr, "generator failed\n"); return 1; }
    }
    if (dump && dump_batch(dump, &p, tasks, n)) { fprintf(stderr, "cannot write %s\n", dump); return 1; }

    /* shard: task k -> GPU (k / batch) mod gpus, gathered into one contiguous array per GPU */
    shard_t *sh = calloc((size_t)gpus, sizeof(*sh));
    size_t *idx = malloc((n ? n : 1) * sizeof(size_t));
    bsw_result *res = malloc((n ? n : 1) * sizeof(*res));
    for (int g = 0; g < gpus; ++g) {
        size_t cnt = 0;
        for (i = 0; i < n; ++i) if ((int)((i / batch) % (size_t)gpus) == g) ++cnt;
        sh[g].gpu = g; sh[g].pack_threads = threads; sh[g].batch = batch; sh[g].p = &p; sh[g].n = cnt;
        sh[g].tasks = malloc((cnt ? cnt : 1) * sizeof(bsw_task)); sh[g].res = malloc((cnt ? cnt : 1) * sizeof(bsw_result));
    }
    {
        size_t *fill = calloc((size_t)gpus, sizeof(size_t));
        for (i = 0; i < n; ++i) { int g = (int)((i / batch) % (size_t)gpus); idx[i] = fill[g]; sh[g].tasks[fill[g]++] = tasks[i]; }
        free(fill);
    }
    pthread_t *th = malloc((size_t)gpus * sizeof(pthread_t));
    double t0 = now();
    for (int g = 0; g < gpus; ++g) pthread_create(&th[g], NULL, shard_main, &sh[g]);
    for (int g = 0; g < gpus; ++g) pthread_join(th[g], NULL);
    double dt = now() - t0;
    for (int g = 0; g < gpus; ++g) if (sh[g].rc != BSW_OK) { fprintf(stderr, "%s\n", sh[g].err); return 1; }
    unsigned long long cells = 0, sum = 0;
    for (i = 0; i < n; ++i) {
        res[i] = sh[(i / batch) % (size_t)gpus].res[idx[i]];
        cells += res[i].left.cells + res[i].right.cells;
        sum = sum * 1315423911ull + (unsigned)res[i].score + ((unsigned long long)(unsigned)res[i].truesc << 20) + (unsigned)res[i].qb * 7u + (unsigned)res[i].re * 13u;
    }
    printf("{\"seeds\": %zu, \"gpus\": %d, \"seconds\": %.4f, \"seeds_per_s\": %.1f, \"gcups_incl_create_pack_pcie\": %.2f, \"result_checksum\": \"%016llx\"}\n",
           n, gpus, dt, n / dt, cells / dt / 1e9, sum);
    return 0;
}
