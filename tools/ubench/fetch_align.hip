// Microbenchmark: what a 64-bit instruction costs when it starts 4 (mod 8) in the instruction stream (gfx950).
// A loop body of N independent 64-bit VALU instructions (v_pk_max_u16 over 8 rotating registers; a second variant with
// 32-bit v_max_u32_e32 for contrast), aligned to 64 bytes and then shifted by SHIFT dwords of s_nop; one or two waves per SIMD.
// Prints cycles per instruction (s_memtime, wave 0 of each workgroup).  The looped two-seeds-per-lane kernel lost 7 % to
// one dword of shift (profiles/r3/fetch_alignment.txt); this isolates the effect.
// hipcc --offload-arch=gfx950 -O3 fetch_align.hip -o fetch_align
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(x) x x x x x x x x
#define BODY64 \
    "v_pk_max_u16 %0, %0, %8\n\tv_pk_max_u16 %1, %1, %8\n\tv_pk_max_u16 %2, %2, %8\n\tv_pk_max_u16 %3, %3, %8\n\t" \
    "v_pk_max_u16 %4, %4, %8\n\tv_pk_max_u16 %5, %5, %8\n\tv_pk_max_u16 %6, %6, %8\n\tv_pk_max_u16 %7, %7, %8\n\t"
#define BODY32 \
    "v_max_u32_e32 %0, %0, %8\n\tv_max_u32_e32 %1, %1, %8\n\tv_max_u32_e32 %2, %2, %8\n\tv_max_u32_e32 %3, %3, %8\n\t" \
    "v_max_u32_e32 %4, %4, %8\n\tv_max_u32_e32 %5, %5, %8\n\tv_max_u32_e32 %6, %6, %8\n\tv_max_u32_e32 %7, %7, %8\n\t"
#define PAD0 ""
#define PAD1 "s_nop 0\n\t"
#define PAD2 "s_nop 0\n\ts_nop 0\n\t"
#define PAD3 "s_nop 0\n\ts_nop 0\n\ts_nop 0\n\t"

template <int SHIFT, bool WIDE, int WPS>
__global__ __launch_bounds__(256, WPS) void k(unsigned long long *out, uint32_t seed, int iters)
{
    uint32_t a = seed, b = seed + 1, c = seed + 2, d = seed + 3, e = seed + 4, f = seed + 5, g = seed + 6, h = seed + 7, y = threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#define RUN(PAD, BODY) asm volatile(".p2align 6\n\t" PAD REP8(REP8(BODY)) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(y))
        if constexpr (WIDE) {
            if constexpr (SHIFT == 0) RUN(PAD0, BODY64); else if constexpr (SHIFT == 1) RUN(PAD1, BODY64);
            else if constexpr (SHIFT == 2) RUN(PAD2, BODY64); else RUN(PAD3, BODY64);
        } else {
            if constexpr (SHIFT == 0) RUN(PAD0, BODY32); else RUN(PAD1, BODY32);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    if (a + b + c + d + e + f + g + h == 0x12345u) out[0] = 0;
}

template <int SHIFT, bool WIDE, int WPS>
static void run(const char *what)
{
    const int blocks = 256 * WPS, iters = 2000, n = 512;          // 512 instructions per iteration
    unsigned long long *d, *hbuf = new unsigned long long[blocks * 4];
    hipMalloc(&d, sizeof(unsigned long long) * blocks * 4);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<SHIFT, WIDE, WPS>), dim3(blocks), dim3(256), 0, 0, d, 1u, iters);
    hipMemcpy(hbuf, d, sizeof(unsigned long long) * blocks * 4, hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < blocks * 4; ++i) s += (double)hbuf[i];
    // s_memtime ticks at 100 MHz on gfx950; report ticks per 1000 instructions (relative numbers are what matters)
    printf("%-8s shift %d dwords, %d wave(s)/SIMD: %.3f memtime ticks per 1000 instructions\n", what, SHIFT, WPS, s / (blocks * 4) / ((double)iters * n) * 1000.0);
    hipFree(d);
    delete[] hbuf;
}

int main()
{
    run<0, true, 1>("64-bit"); run<1, true, 1>("64-bit"); run<2, true, 1>("64-bit"); run<3, true, 1>("64-bit");
    run<0, true, 2>("64-bit"); run<1, true, 2>("64-bit");
    run<0, true, 3>("64-bit"); run<0, true, 4>("64-bit"); run<1, true, 4>("64-bit"); run<0, true, 8>("64-bit");
    run<0, false, 2>("32-bit"); run<0, false, 4>("32-bit");
    run<0, false, 1>("32-bit"); run<1, false, 1>("32-bit");
    return 0;
}
