// VALU issue-rate microbenchmark for gfx950: cycles per wave64 instruction per SIMD for the instruction
// kinds the DP kernels use, at 1..8 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define ITERS 4096
#define UNROLL 16

template <int KIND>
__global__ __launch_bounds__(256) void k(unsigned *out, unsigned seed)
{
    unsigned a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 7u + i * 13u;
    const unsigned c1 = seed | 3u, c2 = seed >> 1;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            unsigned &x = a[u & 7], &y = a[(u + 3) & 7];
            if (KIND == 0) asm volatile("v_add_u32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(y));
            if (KIND == 1) asm volatile("v_max3_i32 %0, %1, %2, %3" : "=v"(x) : "v"(x), "v"(y), "s"(c1));
            if (KIND == 2) asm volatile("v_pk_max_i16 %0, %1, %2" : "=v"(x) : "v"(x), "v"(y));
            if (KIND == 3) asm volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(x) : "v"(x), "v"(y));
            if (KIND == 4) asm volatile("v_lshl_or_b32 %0, %1, 8, %2" : "=v"(x) : "v"(x), "v"(y));
            if (KIND == 5) asm volatile("v_bfe_i32 %0, %1, 5, 1" : "=v"(x) : "v"(y));
            if (KIND == 6) asm volatile("v_cndmask_b32_e32 %0, %1, %2, vcc" : "=v"(x) : "v"(x), "v"(y) : );
            if (KIND == 12) asm volatile("v_cmp_ne_u32_e32 vcc, %0, %1" : : "v"(x), "v"(y) : "vcc");
            if (KIND == 13) asm volatile("v_cmp_ne_u32_e64 s[20:21], %0, %1" : : "v"(x), "v"(y) : "s20", "s21");
            if (KIND == 14) asm volatile("v_cndmask_b32_e64 %0, %1, %2, s[20:21]" : "=v"(x) : "v"(x), "v"(y));
            if (KIND == 15) asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(x) : "v"(x), "v"(y), "s"(c1));
            if (KIND == 16) asm volatile("v_bfi_b32 %0, %1, %2, %3" : "=v"(x) : "v"(x), "v"(y), "v"(a[(u + 5) & 7]));
            if (KIND == 17) asm volatile("v_max_i32_e32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(y));
            if (KIND == 18) asm volatile("v_lshlrev_b32_e32 %0, 8, %1" : "=v"(x) : "v"(y));
            if (KIND == 19) asm volatile("v_and_b32_e32 %0, 0xff00ff, %1" : "=v"(x) : "v"(y));          // VOP2 + 32-bit literal
            if (KIND == 20) asm volatile("v_subrev_u32_e32 %0, %1, %2" : "=v"(x) : "s"(c1), "v"(y));     // SGPR operand
            if (KIND == 21) asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(x) : "v"(x), "v"(y), "v"(a[(u + 5) & 7]));
            if (KIND == 22) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(x) : "v"(x), "v"(y), "s"(c1));
            if (KIND == 23) asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(x) : "v"(x), "v"(y), "s"(c1));
            if (KIND == 24) asm volatile("v_max_i32_e64 %0, %1, %2" : "=v"(x) : "v"(x), "v"(y));        // VOP3 encoding of a 2-input op
            if (KIND == 25) asm volatile("v_min_u32_e32 %0, 37, %1" : "=v"(x) : "v"(y));
            if (KIND == 7) asm volatile("v_pk_add_u16 %0, %1, %2" : "=v"(x) : "v"(x), "v"(y));
            if (KIND == 8) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "=v"(x) : "v"(y));
            if (KIND == 9) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[0]) : "v"(c2));   // fully dependent chain
            if (KIND == 10) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*(double *)&a[(u & 3) * 2]) : "v"(*(double *)&a[0]), "v"(*(double *)&a[2]));
            if (KIND == 11) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(y), "v"(c2));
        }
    }
    unsigned r = 0;
    for (int i = 0; i < 8; ++i) r ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int KIND>
static void run(const char *name, unsigned *d_out, int cus, double ghz)
{
    for (int wps = 2; wps <= 8; wps *= 2) {
        const int blocks = cus * wps;                 // 256-thread blocks = 4 waves = 1 wave per SIMD each
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, 12345u);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, 12345u);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_simd = (double)ITERS * UNROLL * wps;
        printf("{\"op\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_instr_per_simd\": %.2f}\n", name, wps, ms * 1e-3 * ghz * 1e9 / instr_per_simd);
    }
}

int main()
{
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount; const double ghz = pr.clockRate * 1e-6;
    unsigned *d_out; hipMalloc(&d_out, (size_t)cus * 8 * 256 * 4);
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_ghz\": %.3f}\n", pr.gcnArchName, cus, ghz);
    run<0>("v_add_u32", d_out, cus, ghz);
    run<1>("v_max3_i32", d_out, cus, ghz);
    run<2>("v_pk_max_i16", d_out, cus, ghz);
    run<7>("v_pk_add_u16", d_out, cus, ghz);
    run<3>("v_add_u32_sdwa", d_out, cus, ghz);
    run<4>("v_lshl_or_b32", d_out, cus, ghz);
    run<5>("v_bfe_i32", d_out, cus, ghz);
    run<6>("v_cndmask_b32", d_out, cus, ghz);
    run<8>("v_mov_b32_dpp", d_out, cus, ghz);
    run<12>("v_cmp_ne_u32_e32 (vcc)", d_out, cus, ghz);
    run<13>("v_cmp_ne_u32_e64 (sgpr pair)", d_out, cus, ghz);
    run<14>("v_cndmask_b32_e64 (sgpr mask)", d_out, cus, ghz);
    run<15>("v_and_or_b32", d_out, cus, ghz);
    run<16>("v_bfi_b32", d_out, cus, ghz);
    run<17>("v_max_i32_e32", d_out, cus, ghz);
    run<24>("v_max_i32_e64", d_out, cus, ghz);
    run<18>("v_lshlrev_b32_e32", d_out, cus, ghz);
    run<19>("v_and_b32_e32 + literal", d_out, cus, ghz);
    run<25>("v_min_u32_e32 inline const", d_out, cus, ghz);
    run<20>("v_subrev_u32_e32 sgpr", d_out, cus, ghz);
    run<21>("v_mad_u32_u24", d_out, cus, ghz);
    run<22>("v_perm_b32", d_out, cus, ghz);
    run<23>("v_add3_u32", d_out, cus, ghz);
    run<9>("v_add_u32 dependent chain", d_out, cus, ghz);
    run<11>("v_fma_f32", d_out, cus, ghz);
    run<10>("v_pk_fma_f32", d_out, cus, ghz);
    return 0;
}
