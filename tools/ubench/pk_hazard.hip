// Is the one wait state the compiler puts between a VOP3P (packed 16-bit) instruction and a dependent VALU read real on
// gfx950, and what does it cost?  Three kernels run the same dependent chain of v_pk_max_u16 / v_pk_sub_u16 clamp /
// v_and_b32: (A) one hand-written asm block without s_nop, (B) the same block with s_nop 0 after every instruction,
// (C) separate asm statements scheduled by the compiler (it inserts the s_nops).  Results are compared with the CPU.
// hipcc --offload-arch=gfx950 -O3 pk_hazard.hip -o pk_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHAIN_A(N) \
    "v_pk_max_u16 %0, %0, %1\n" N \
    "v_pk_sub_u16 %0, %0, %2 clamp\n" N \
    "v_pk_add_u16 %0, %0, %3\n" N \
    "v_and_b32 %0, %0, %4\n" N \
    "v_pk_max_u16 %0, %0, %2\n" N \
    "v_pk_mad_u16 %0, %0, %5, %1\n" N \
    "v_pk_lshrrev_b16 %0, 1, %0 op_sel_hi:[0,1]\n" N \
    "v_pk_min_u16 %0, %0, %4\n" N

template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t *out, const uint32_t *in, int iters, long long *cyc)
{
    uint32_t x = in[threadIdx.x], a = in[256 + threadIdx.x], b = in[512 + threadIdx.x], c = in[768 + threadIdx.x];
    const uint32_t m = 0x3fff3fffu, one = 0x00010001u;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) asm volatile(CHAIN_A("") CHAIN_A("") CHAIN_A("") CHAIN_A("") : "+v"(x) : "v"(a), "v"(b), "v"(c), "v"(m), "v"(one));
        else if (MODE == 1) asm volatile(CHAIN_A("s_nop 0\n") CHAIN_A("s_nop 0\n") CHAIN_A("s_nop 0\n") CHAIN_A("s_nop 0\n") : "+v"(x) : "v"(a), "v"(b), "v"(c), "v"(m), "v"(one));
        else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(x) : "v"(a));
                asm volatile("v_pk_sub_u16 %0, %0, %1 clamp" : "+v"(x) : "v"(b));
                asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(x) : "v"(c));
                x &= m;
                asm volatile("" : "+v"(x));
                asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(x) : "v"(b));
                asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(x) : "v"(one), "v"(a));
                asm volatile("v_pk_lshrrev_b16 %0, 1, %0 op_sel_hi:[0,1]" : "+v"(x));
                asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(x) : "v"(m));
            }
        }
    }
    const long long t1 = clock64();
    out[blockIdx.x * 256 + threadIdx.x] = x;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

static uint32_t pk(uint32_t a, uint32_t b, int op)
{
    uint32_t r = 0;
    for (int h = 0; h < 2; ++h) {
        uint32_t x = (a >> (16 * h)) & 0xffff, y = (b >> (16 * h)) & 0xffff, z;
        switch (op) {
        case 0: z = x > y ? x : y; break;
        case 1: z = x > y ? x - y : 0; break;
        case 2: z = (x + y) & 0xffff; break;
        case 3: z = x < y ? x : y; break;
        default: z = x >> 1; break;
        }
        r |= z << (16 * h);
    }
    return r;
}

int main()
{
    std::vector<uint32_t> in(1024);
    uint32_t s = 12345;
    for (auto &v : in) { s = s * 1664525u + 1013904223u; v = s & 0x1fff1fffu; }
    uint32_t *din, *dout; long long *dc;
    hipMalloc(&din, 4096); hipMalloc(&dout, 4 * 256 * 2048); hipMalloc(&dc, 8);
    hipMemcpy(din, in.data(), 4096, hipMemcpyHostToDevice);
    const int iters = 20000;
    std::vector<uint32_t> want(256);
    for (int t = 0; t < 256; ++t) {
        uint32_t x = in[t], a = in[256 + t], b = in[512 + t], c = in[768 + t];
        for (int i = 0; i < iters * 4; ++i) {
            x = pk(x, a, 0); x = pk(x, b, 1); x = pk(x, c, 2); x &= 0x3fff3fffu; x = pk(x, b, 0);
            x = pk(x, a, 2);               /* mad x*1 + a */
            x = pk(x, 0, 4); x = pk(x, 0x3fff3fffu, 3);
        }
        want[t] = x;
    }
    const char *names[3] = {"asm block, no s_nop", "asm block, s_nop 0 after every op", "compiler-scheduled (inserts s_nop)"};
    for (int wps = 1; wps <= 8; wps *= 2)
        for (int mode = 0; mode < 3; ++mode) {
            const int blocks = 256 * wps;
            std::vector<uint32_t> got(256);
            long long cyc = 0;
            float ms = 0;
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                (void)hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, dout, din, iters, dc);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, dout, din, iters, dc);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, dout, din, iters, dc);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                (void)hipEventElapsedTime(&ms, e0, e1);
            }
            hipMemcpy(got.data(), dout, 1024, hipMemcpyDeviceToHost);
            hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost);
            int bad = 0;
            for (int t = 0; t < 256; ++t) bad += got[t] != want[t];
            printf("%d wave(s)/SIMD  %-38s  %6.2f cycles (2.4 GHz) per VALU op per SIMD   [%.3f ms]   mismatching lanes %d/256\n", wps, names[mode],
                   ms * 1e-3 * 2.4e9 / (iters * 32.0 * wps), ms, bad);
            fflush(stdout);
        }
    return 0;
}
