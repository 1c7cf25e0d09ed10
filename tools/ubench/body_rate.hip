// Microbenchmark: the generated dense block body (bsw_lane2_body_asm.inc, 126 VALU instructions per 8 pair-columns) back to
// back, 17 blocks per "row" on 136 row registers, at one and two waves per SIMD — what the body's own schedule allows, without
// the kernel's dispatch, reductions and tails.  Compare with the issue ceiling of independent instructions
// (fetch_align.hip: 4.18 / 3.05 cycles per instruction and SIMD at 1 / 2 waves).
// hipcc --offload-arch=gfx950 -O3 -I../../bwa-mem-sw_amd/csrc -I../../include body_rate.hip -o body_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define L2_STAMP(k) ((void)0)
#define BSW_L2_ASM_BODY 1
#ifdef GRID
#define BSW_L2_GRID 1
#endif
#include "bsw_device.h"
#include "bsw_lane2_core.h"

using namespace bsw::l2;

template <int WPS, bool EDGE>
__global__ __launch_bounds__(256, WPS) void k(unsigned long long *out, uint32_t seed, int rows)
{
    using L = lane2<17, false, true>;
    consts c;
    c.a = 1; c.pb = 4; c.pn = 1; c.o_del = 6; c.e_del = 1; c.oe_ins = 7; c.e_ins = 1; c.zdrop = 100;
    fill_packed_consts(c);
    uint32_t Pr[136];
    sfor<136>([&](auto ji) { Pr[decltype(ji)::value] = (seed + threadIdx.x * 7u + decltype(ji)::value) & 0x0f000f00u; });
    uint32_t h1 = 0, f = 0, acc = 0, W = seed * 0x01010101u, Bv = 0x04000400u, END = 0x00880088u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < rows; ++i) {
        sfor<17>([&](auto bi) {
            constexpr int b = decltype(bi)::value;
            uint32_t T[8], mk, nz;
            sfor<8>([&](auto ci) { T[decltype(ci)::value] = Pr[8 * b + decltype(ci)::value]; });
            L::template block8<EDGE, false>(T, W, 0u, Bv, 0u, c, END, 0xffffffffu, h1, f, mk, nz);
            sfor<8>([&](auto ci) { Pr[8 * b + decltype(ci)::value] = T[decltype(ci)::value]; });
            acc ^= mk ^ nz;
        });
        asm volatile("" : "+v"(W));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    uint32_t x = acc ^ h1 ^ f;
    sfor<136>([&](auto ji) { x ^= Pr[decltype(ji)::value]; });
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    if (x == 0x12345u) out[0] = 0;
}

template <int WPS, bool EDGE>
static void run(int ninst)
{
    const int blocks = 256 * WPS, rows = 300;
    unsigned long long *d, *h = new unsigned long long[blocks * 4];
    (void)hipMalloc(&d, sizeof(unsigned long long) * blocks * 4);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<WPS, EDGE>), dim3(blocks), dim3(256), 0, 0, d, 3u, rows);
    (void)hipMemcpy(h, d, sizeof(unsigned long long) * blocks * 4, hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < blocks * 4; ++i) s += (double)h[i];
    const double per_wave = s / (blocks * 4) / ((double)rows * 17 * ninst);
    printf("%s body (%d instructions per block), %d wave(s)/SIMD: %.2f cycles per instruction and wave = %.2f per SIMD\n",
           EDGE ? "edge " : "dense", ninst, WPS, per_wave, per_wave / WPS);
    (void)hipFree(d);
    delete[] h;
}

int main()
{
    run<1, false>(128); run<2, false>(128); run<1, true>(183); run<2, true>(183);
    return 0;
}
