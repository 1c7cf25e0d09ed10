// Microbenchmark of the two-seeds-per-lane ("pair") dense cell body of the lane kernel at 1 / 2 waves per SIMD:
// 136 eh[] columns in 136 VGPRs (each = two seeds' {e:8,h:8}), 257 DP rows, every column dense.
// Prints cycles per pair-cell per SIMD; the real kernel's budget is ~19 instructions x ~4.7 cycles at 2 waves/SIMD.
// hipcc --offload-arch=gfx950 -O3 pair_body.hip -o pair_body
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <utility>

#define PK2(op, d, a, b) asm(op " %0, %1, %2" : "=v"(d) : "v"(a), "v"(b))
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) { uint32_t d; PK2("v_pk_max_u16", d, a, b); return d; }
__device__ __forceinline__ uint32_t pk_min(uint32_t a, uint32_t b) { uint32_t d; PK2("v_pk_min_u16", d, a, b); return d; }
__device__ __forceinline__ uint32_t pk_subs(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ uint32_t pk_mad(uint32_t a, uint32_t b, uint32_t c) { uint32_t d; asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

struct consts { uint32_t AB, Bv, OE, ED, ONE, C256; };

template <int NP, int WPS, int MODE>
__global__ __launch_bounds__(256, WPS) void k(uint32_t *out, const uint32_t *in, int rows, consts c)
{
    uint32_t Pr[NP];
    static_for<NP>([&](auto ci) { constexpr int J = decltype(ci)::value; Pr[J] = in[(threadIdx.x + J * 64) & 4095] & 0x0f0f0f0fu; });
    uint32_t mkacc = 0, nzall = 0;
    asm volatile("" : "+v"(c.AB), "+v"(c.Bv), "+v"(c.OE), "+v"(c.ED), "+v"(c.ONE), "+v"(c.C256));
    for (int i = 0; i < rows; ++i) {
        uint32_t W[(NP + 15) / 16];
#pragma unroll
        for (int w = 0; w < (NP + 15) / 16; ++w) W[w] = in[(threadIdx.x * 3 + i * 17 + w * 5) & 4095];
        uint32_t f = 0, h1 = (uint32_t)i & 0x00030003u, mk = 0;
        uint32_t Jc = 0;
        asm volatile("" : "+v"(Jc));
        static_for<NP>([&](auto ci) {
            constexpr int J = decltype(ci)::value;
            uint32_t &P = Pr[J];
            uint32_t t = W[J >> 4];
            if ((J & 15) != 0) t >>= (J & 15);
            t &= 0x00010001u;
            const uint32_t hd = P & 0x00ff00ffu;
            const uint32_t e = (P >> 8) & 0x00ff00ffu;
            uint32_t M = pk_subs(pk_mad(t, c.AB, hd), c.Bv);
            uint32_t h = pk_max(pk_max(M, e), f);
            uint32_t key = MODE == 0 ? pk_mad(h, c.C256, Jc) : ((h << 8) | Jc);
            mk = pk_max(mk, key);
            const uint32_t tD = pk_subs(h, c.OE);
            const uint32_t en = pk_max(pk_subs(e, c.ED), tD);
            f = pk_max(pk_subs(f, c.ED), tD);
            const uint32_t np = (en << 8) | h1;
            uint32_t nzb = pk_min(np, c.ONE);
            nzall = pk_mad(nzb, c.ONE, nzall);          // stands in for the bit-placing pk_mad (same cost)
            h1 = h;
            P = np;
            Jc += 0x00010001u;
        });
        mkacc ^= mk ^ f ^ h1;
    }
    uint32_t r = mkacc ^ nzall;
    static_for<NP>([&](auto ci) { r ^= Pr[decltype(ci)::value]; });
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int NP, int WPS, int MODE>
static void run(const char *name, uint32_t *d_out, const uint32_t *d_in, int cus, double ghz)
{
    const int rows = 257;
    consts c{0x00050005u, 0x00040004u, 0x00070007u, 0x00010001u, 0x00010001u, 0x01000100u};
    const int blocks = cus * WPS;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NP, WPS, MODE>), dim3(blocks), dim3(256), 0, 0, d_out, d_in, rows, c);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NP, WPS, MODE>), dim3(blocks), dim3(256), 0, 0, d_out, d_in, rows, c);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double pair_cells_per_simd = (double)rows * NP * WPS;       // one wave per SIMD per WPS
    const double cyc = ms * 1e-3 * ghz * 1e9 / pair_cells_per_simd;
    printf("%-44s NP=%d waves/SIMD=%d: %.3f ms, %.1f cycles per pair-cell per SIMD = %.1f per cell -> %.0f GCUPS if every cell were dense\n",
           name, NP, WPS, ms, cyc, cyc / 2, 2.0 * 64 / cyc * cus * 4 * ghz);
    fflush(stdout);
}

int main()
{
    hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount; const double ghz = pr.clockRate * 1e-6;
    uint32_t *d_out, *d_in;
    (void)hipMalloc(&d_out, (size_t)cus * 8 * 256 * 4);
    (void)hipMalloc(&d_in, 4096 * 4);
    (void)hipMemset(d_in, 0x11, 4096 * 4);
    run<136, 2, 0>("pair dense body (key = pk_mad)", d_out, d_in, cus, ghz);
    run<136, 2, 1>("pair dense body (key = lshl_or)", d_out, d_in, cus, ghz);
    run<136, 1, 0>("pair dense body, 1 wave/SIMD", d_out, d_in, cus, ghz);
    run<72, 4, 0>("pair dense body, 72 columns, 4 waves/SIMD", d_out, d_in, cus, ghz);
    run<72, 2, 0>("pair dense body, 72 columns, 2 waves/SIMD", d_out, d_in, cus, ghz);
    run<104, 3, 0>("pair dense body, 104 columns, 3 waves/SIMD", d_out, d_in, cus, ghz);
    run<232, 1, 0>("pair dense body, 232 columns, 1 wave/SIMD", d_out, d_in, cus, ghz);
    run<232, 1, 1>("pair dense body, 232 columns, 1 wave/SIMD, lshl_or key", d_out, d_in, cus, ghz);
    run<200, 1, 0>("pair dense body, 200 columns, 1 wave/SIMD", d_out, d_in, cus, ghz);
    return 0;
}
