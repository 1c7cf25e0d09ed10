/*
 * bsw_bench.c — "bsw-bench": a plain-C host over the C ABI (what INTEGRATION.md describes).
 *
 * Mirrors the reference host's command line where it applies (reference README.md:29-36:
 * `bwa --target=ASE|Direct mem -t N -b BATCH ...`) with bwa mem's scoring flags:
 *
 *   bsw-bench [--target=hip] [--gpus G | --devices 0,1,..] [-t gather_threads] [-b batch_seeds] [-n seeds] [-l read_len]
 *             [-A a] [-B b] [-O o[,o_ins]] [-E e[,e_ins]] [-L clip] [-w band] [-d zdrop] [--variant=H|M] [--reps R] [--pageable]
 *             [--packed] [--dump FILE | --load FILE]
 *
 * --target=cpu is refused: the library has no CPU path (the CPU oracle lives under oracle/ and is test-only).
 * --gpus G / --devices: ONE context drives all the GPUs; the library sends chunk k of the seed pool to device
 * k mod G, like the reference's round-robin over its 4 PE arrays (batch_manager.v:343-348) — no inter-GPU traffic.
 * Sequences and results live in bsw_host_alloc memory (DMA direct) unless --pageable.
 * --packed: the sequences are handed over 4-bit packed (bsw_pack_tasks once, then bsw_submit_packed): what a host does
 * that keeps its reads packed, as the reference's host does on its link.  -O 6,4 / -E 1,2: deletion,insertion penalties
 * (bwa's own -O / -E syntax).
 * --dump / --load write / read a self-contained task batch (params + seeds + sequences) for reproducible runs.
 * Prints one JSON line; result_checksum folds score, truesc, qb and re of every seed in task order.
 *
 * Build: make -C bwa-mem-sw_amd/csrc bsw-bench
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "bwa_sw_mi355.h"

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + (double)t.tv_nsec * 1e-9; }

/* dump format: magic, bsw_params, n, per seed {4 lengths, h0, init_score, qbeg, tag, wlim_l, wlim_r}, then all sequences lq|lt|rq|rt */
#define DUMP_MAGIC 0x3248435441425753ull                                 /* "SWBATCH2" */
static int dump_batch(const char *path, const bsw_params *p, const bsw_task *t, size_t n)
{
    FILE *f = fopen(path, "wb");
    size_t i;
    const uint64_t magic = DUMP_MAGIC, nn = n;
    if (!f) return -1;
    fwrite(&magic, 8, 1, f); fwrite(p, sizeof(*p), 1, f); fwrite(&nn, 8, 1, f);
    for (i = 0; i < n; ++i) {
        int32_t h[10] = {t[i].lqlen, t[i].ltlen, t[i].rqlen, t[i].rtlen, t[i].h0, t[i].init_score, t[i].qbeg, (int32_t)t[i].tag, t[i].wlim_l, t[i].wlim_r};
        fwrite(h, sizeof(h), 1, f);
    }
    for (i = 0; i < n; ++i) {
        if (t[i].lqlen) { fwrite(t[i].lquery, 1, (size_t)t[i].lqlen, f); fwrite(t[i].ltarget, 1, (size_t)t[i].ltlen, f); }
        if (t[i].rqlen) { fwrite(t[i].rquery, 1, (size_t)t[i].rqlen, f); fwrite(t[i].rtarget, 1, (size_t)t[i].rtlen, f); }
    }
    return fclose(f);
}

static void *arena_alloc(size_t bytes, int pageable) { return pageable ? malloc(bytes) : bsw_host_alloc(bytes); }

static int load_batch(const char *path, int pageable, bsw_params *p, bsw_task **tasks, uint8_t **arena, size_t *n)
{
    FILE *f = fopen(path, "rb");
    uint64_t magic = 0, nn = 0;
    size_t total = 0, off = 0, i;
    bsw_task *t;
    uint8_t *a;
    if (!f) return -1;
    if (fread(&magic, 8, 1, f) != 1 || magic != DUMP_MAGIC || fread(p, sizeof(*p), 1, f) != 1 || fread(&nn, 8, 1, f) != 1) { fclose(f); return -2; }
    t = calloc(nn ? nn : 1, sizeof(*t));
    for (i = 0; i < nn; ++i) {
        int32_t h[10];
        if (fread(h, sizeof(h), 1, f) != 1) { fclose(f); return -3; }
        t[i].lqlen = h[0]; t[i].ltlen = h[1]; t[i].rqlen = h[2]; t[i].rtlen = h[3];
        t[i].h0 = h[4]; t[i].init_score = h[5]; t[i].qbeg = h[6]; t[i].tag = (uint32_t)h[7]; t[i].wlim_l = h[8]; t[i].wlim_r = h[9];
        total += (size_t)(h[0] ? h[0] + h[1] : 0) + (size_t)(h[2] ? h[2] + h[3] : 0);
    }
    a = arena_alloc(total + 64, pageable);
    if (!a || fread(a, 1, total, f) != total) { fclose(f); return -4; }
    for (i = 0; i < nn; ++i) {
        if (t[i].lqlen) { t[i].lquery = a + off; off += (size_t)t[i].lqlen; t[i].ltarget = a + off; off += (size_t)t[i].ltlen; }
        if (t[i].rqlen) { t[i].rquery = a + off; off += (size_t)t[i].rqlen; t[i].rtarget = a + off; off += (size_t)t[i].rtlen; }
    }
    fclose(f);
    *tasks = t; *arena = a; *n = nn;
    return 0;
}

int main(int argc, char **argv)
{
    int packed = 0, o_ins = -1, e_ins = -1;
    int a = 1, b = 4, o = 6, e = 1, clip = 5, w = 100, zdrop = 100, read_len = 150, variant = BSW_VARIANT_H, threads = 4, reps = 1, pageable = 0;
    int ndev = 1, devs[BSW_MAX_DEVICES] = {0}, k, r, c;
    size_t n = 200000, batch = 65536, i;
    const char *dump = NULL, *load = NULL;
    for (k = 1; k < argc; ++k) {
        const char *f = argv[k], *v = k + 1 < argc ? argv[k + 1] : "";
        if (!strcmp(f, "-A")) a = atoi(v), ++k; else if (!strcmp(f, "-B")) b = atoi(v), ++k;
        else if (!strcmp(f, "-O")) { o = atoi(v); if (strchr(v, ',')) o_ins = atoi(strchr(v, ',') + 1); ++k; }
        else if (!strcmp(f, "-E")) { e = atoi(v); if (strchr(v, ',')) e_ins = atoi(strchr(v, ',') + 1); ++k; }
        else if (!strcmp(f, "--packed")) packed = 1;
        else if (!strcmp(f, "-L")) clip = atoi(v), ++k; else if (!strcmp(f, "-w")) w = atoi(v), ++k;
        else if (!strcmp(f, "-d")) zdrop = atoi(v), ++k; else if (!strcmp(f, "-b")) batch = (size_t)atol(v), ++k;
        else if (!strcmp(f, "-n")) n = (size_t)atol(v), ++k; else if (!strcmp(f, "-l")) read_len = atoi(v), ++k;
        else if (!strcmp(f, "-t")) threads = atoi(v), ++k; else if (!strcmp(f, "--reps")) reps = atoi(v), ++k;
        else if (!strcmp(f, "--gpus")) { ndev = atoi(v); ++k; if (ndev < 1 || ndev > BSW_MAX_DEVICES) return 2; for (c = 0; c < ndev; ++c) devs[c] = c; }
        else if (!strcmp(f, "--devices")) {
            char buf[256], *tok;
            ndev = 0; snprintf(buf, sizeof(buf), "%s", v); ++k;
            for (tok = strtok(buf, ","); tok && ndev < BSW_MAX_DEVICES; tok = strtok(NULL, ",")) devs[ndev++] = atoi(tok);
            if (ndev < 1) return 2;
        }
        else if (!strcmp(f, "--dump")) dump = v, ++k; else if (!strcmp(f, "--load")) load = v, ++k;
        else if (!strcmp(f, "--pageable")) pageable = 1;
        else if (!strcmp(f, "--variant=M")) variant = BSW_VARIANT_M; else if (!strcmp(f, "--variant=H")) variant = BSW_VARIANT_H;
        else if (!strcmp(f, "--target=hip")) {}
        else if (!strncmp(f, "--target=", 9)) { fprintf(stderr, "%s: only --target=hip exists; this library has no CPU path\n", f); return 2; }
        else { fprintf(stderr, "unknown flag %s\n", f); return 2; }
    }
    if (batch < 1 || reps < 1) return 2;
    bsw_params p; bsw_default_params(&p);
    bsw_task *tasks = NULL; uint8_t *arena = NULL;
    if (load) {
        int rc = load_batch(load, pageable, &p, &tasks, &arena, &n);
        if (rc) { fprintf(stderr, "cannot load %s (%d)\n", load, rc); return 1; }
    } else {
        bsw_synth_spec sp;
        size_t cap;
        for (r = 0; r < 5; ++r) for (c = 0; c < 5; ++c) p.mat[r * 5 + c] = (int8_t)((r == 4 || c == 4) ? -1 : (r == c ? a : -b));
        p.o_del = o; p.o_ins = o_ins >= 0 ? o_ins : o; p.e_del = e; p.e_ins = e_ins >= 1 ? e_ins : e; p.pen_clip5 = p.pen_clip3 = clip; p.w = w; p.zdrop = zdrop; p.variant = variant;
        memset(&sp, 0, sizeof(sp));
        sp.seed = 1; sp.read_len = read_len; sp.seed_len_min = 19; sp.seed_len_max = 60; sp.seed_at_start = 0;
        sp.sub_rate = 0.01; sp.indel_rate = 0.001; sp.junk_frac = 0.05; sp.a = a; sp.w = w; sp.o = o; sp.e = e;
        cap = bsw_synth_arena_bound(&sp, n);
        arena = arena_alloc(cap, pageable); tasks = malloc((n ? n : 1) * sizeof(*tasks));
        if (!arena || !tasks || bsw_synth_generate(&sp, n, tasks, arena, cap) < 0) { fprintf(stderr, "generator failed (no GPU for pinned memory? try --pageable)\n"); return 1; }
    }
    if (dump && dump_batch(dump, &p, tasks, n)) { fprintf(stderr, "cannot write %s\n", dump); return 1; }

    bsw_config cfg; bsw_default_config(&cfg);
    cfg.chunk_tasks = batch; cfg.pack_threads = threads; cfg.n_devices = ndev;
    for (c = 0; c < ndev; ++c) cfg.devices[c] = devs[c];
    bsw_ctx *ctx = NULL;
    int rc = bsw_create(&cfg, &ctx);
    if (rc != BSW_OK) { fprintf(stderr, "bsw_create failed (%d): no CPU path exists\n", rc); return 1; }
    bsw_result *res = arena_alloc((n ? n : 1) * sizeof(*res), pageable);
    if (!res) { fprintf(stderr, "out of memory\n"); return 1; }
    memset(res, 0, (n ? n : 1) * sizeof(*res));
    if (packed) {                                                   /* pack once: the timed passes hand packed words over */
        const size_t cap = bsw_pack_tasks_bound(tasks, n);
        uint64_t *parena = arena_alloc(cap, pageable);
        bsw_task *pt = malloc((n ? n : 1) * sizeof(*pt));
        if (!parena || !pt || bsw_pack_tasks(tasks, n, parena, cap, pt) < 0) { fprintf(stderr, "packing failed\n"); return 1; }
        tasks = pt;
    }
    double best = 1e30;
    for (r = 0; r < reps + (reps > 1); ++r) {                       /* with --reps > 1 the first pass is a warm-up */
        const double t0 = now();
        rc = packed ? bsw_submit_packed(ctx, &p, tasks, n, res) : bsw_submit(ctx, &p, tasks, n, res);   /* = ring CSR_REQ_PEARRAY */
        if (rc == BSW_OK) rc = bsw_wait(ctx);                       /* = poll the DSM busy bits */
        if (rc != BSW_OK) { fprintf(stderr, "%s\n", bsw_last_error(ctx)); return 1; }
        if ((r > 0 || reps == 1) && now() - t0 < best) best = now() - t0;
    }
    bsw_destroy(ctx);
    unsigned long long cells = 0, sum = 0;
    for (i = 0; i < n; ++i) {
        cells += res[i].left.cells + res[i].right.cells;
        sum = sum * 1315423911ull + (unsigned)res[i].score + ((unsigned long long)(unsigned)res[i].truesc << 20) + (unsigned)res[i].qb * 7u + (unsigned)res[i].re * 13u;
    }
    printf("{\"seeds\": %zu, \"packed_input\": %d, \"gpus\": %d, \"seconds\": %.5f, \"seeds_per_s\": %.1f, \"gcups_pcie_inclusive\": %.2f, \"cells\": %llu, \"result_checksum\": \"%016llx\"}\n",
           n, packed, ndev, best, (double)n / best, (double)cells / best / 1e9, cells, sum);
    return 0;
}
