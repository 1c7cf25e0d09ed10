/* scalar_rate.c — calls/s and latency of the drop-in ksw_extend2 from T threads (what `bwa mem -t T` does).
 * gcc -O2 -Iinclude tools/scalar_rate.c -Lbwa-mem-sw_amd -lbwasw_mi355 -lpthread -Wl,-rpath,$PWD/bwa-mem-sw_amd -o tools/scalar_rate */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "bwa_sw_mi355.h"
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + (double)t.tv_nsec * 1e-9; }
typedef struct { int calls, qlen, tlen; unsigned seed; long long sum; } arg_t;
static int8_t mat[25];
static void *work(void *p)
{
    arg_t *a = (arg_t *)p;
    uint8_t *q = malloc((size_t)a->qlen), *t = malloc((size_t)a->tlen);
    unsigned s = a->seed;
    for (int c = 0; c < a->calls; ++c) {
        for (int i = 0; i < a->tlen; ++i) { s = s * 1664525u + 1013904223u; t[i] = (uint8_t)(s >> 30); }
        for (int i = 0; i < a->qlen; ++i) { s = s * 1664525u + 1013904223u; q[i] = (s >> 8) % 50 ? t[i] : (uint8_t)(s >> 30); }
        int qle, tle, gtle, gscore, moff;
        a->sum += ksw_extend2(a->qlen, q, a->tlen, t, 5, mat, 6, 1, 6, 1, 100, 5, 100, 19, &qle, &tle, &gtle, &gscore, &moff);
    }
    free(q); free(t);
    return NULL;
}
int main(int argc, char **argv)
{
    int threads = argc > 1 ? atoi(argv[1]) : 16, calls = argc > 2 ? atoi(argv[2]) : 2000;
    for (int i = 0; i < 5; ++i) for (int j = 0; j < 5; ++j) mat[i * 5 + j] = (int8_t)((i == 4 || j == 4) ? -1 : (i == j ? 1 : -4));
    pthread_t th[256]; arg_t a[256];
    { arg_t w = {50, 131, 257, 1u, 0}; work(&w); }                 /* warm up: context creation, code load */
    uint64_t c0, t0; bsw_scalar_stats(&c0, &t0);
    double s0 = now();
    for (int k = 0; k < threads; ++k) { a[k] = (arg_t){calls, 131, 257, 77u + (unsigned)k, 0}; pthread_create(&th[k], NULL, work, &a[k]); }
    long long sum = 0;
    for (int k = 0; k < threads; ++k) { pthread_join(th[k], NULL); sum += a[k].sum; }
    double dt = now() - s0;
    uint64_t c1, t1; bsw_scalar_stats(&c1, &t1);
    printf("{\"threads\": %d, \"calls\": %d, \"seconds\": %.4f, \"calls_per_s\": %.0f, \"us_per_call_per_thread\": %.1f, \"device_round_trips\": %llu, \"calls_per_round_trip\": %.2f, \"score_sum\": %lld}\n",
           threads, threads * calls, dt, threads * calls / dt, dt / calls * 1e6, (unsigned long long)(t1 - t0), (double)(c1 - c0) / (double)(t1 - t0), sum);
    return 0;
}
