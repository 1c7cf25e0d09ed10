// Second VALU microbenchmark: identical operand pattern (x = op(x, y), all VGPR, 8 independent chains) for plain
// VOP2/VOP1 integer opcodes, to separate "opcode class" from "encoding".  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 4096
#define UNROLL 16
#define OPS(X) \
  X(0, "v_add_u32_e32 %0, %1, %2") X(1, "v_sub_u32_e32 %0, %1, %2") X(2, "v_and_b32_e32 %0, %1, %2") \
  X(3, "v_or_b32_e32 %0, %1, %2") X(4, "v_xor_b32_e32 %0, %1, %2") X(5, "v_max_i32_e32 %0, %1, %2") \
  X(6, "v_min_u32_e32 %0, %1, %2") X(7, "v_max_u32_e32 %0, %1, %2") X(8, "v_lshlrev_b32_e32 %0, %1, %2") \
  X(9, "v_lshrrev_b32_e32 %0, %1, %2") X(10, "v_ashrrev_i32_e32 %0, %1, %2") X(11, "v_mul_u32_u24_e32 %0, %1, %2") \
  X(12, "v_cndmask_b32_e32 %0, %1, %2, vcc") X(13, "v_add_f32_e32 %0, %1, %2") X(14, "v_mul_f32_e32 %0, %1, %2") \
  X(15, "v_max_f32_e32 %0, %1, %2") X(16, "v_subrev_u32_e32 %0, %1, %2") X(17, "v_mul_i32_i24_e32 %0, %1, %2") \
  X(18, "v_max_i16_e32 %0, %1, %2") X(19, "v_add_u16_e32 %0, %1, %2") X(20, "v_max_u16_e32 %0, %1, %2") \
  X(21, "v_sub_u16_e32 %0, %1, %2") X(22, "v_xnor_b32_e32 %0, %1, %2") X(23, "v_min_i32_e32 %0, %1, %2")
static const char *names[] = {
#define X(i, s) s,
  OPS(X)
#undef X
};
template <int KIND>
__global__ __launch_bounds__(256) void k(unsigned *out, unsigned seed)
{
    unsigned a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 7u + i * 13u;
    asm volatile("v_cmp_lt_u32_e32 vcc, %0, %1" : : "v"(a[0]), "v"(a[1]) : "vcc");
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            unsigned &x = a[u & 7], &y = a[(u + 3) & 7];
#define X(i, s) if (KIND == i) asm volatile(s : "=v"(x) : "v"(x), "v"(y));
            OPS(X)
#undef X
        }
    }
    unsigned r = 0;
    for (int i = 0; i < 8; ++i) r ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int KIND>
static void run(unsigned *d_out, int cus, double ghz)
{
    printf("%-36s", names[KIND]);
    for (int wps = 2; wps <= 8; wps *= 2) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(cus * wps), dim3(256), 0, 0, d_out, 12345u);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(cus * wps), dim3(256), 0, 0, d_out, 12345u);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf(" w%d: %.2f", wps, ms * 1e-3 * ghz * 1e9 / ((double)ITERS * UNROLL * wps));
    }
    printf("\n");
}
template <int K> static void all(unsigned *d, int cus, double ghz) { run<K>(d, cus, ghz); if constexpr (K + 1 < 24) all<K + 1>(d, cus, ghz); }
int main()
{
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    unsigned *d; hipMalloc(&d, (size_t)pr.multiProcessorCount * 8 * 256 * 4);
    printf("cycles per wave64 instruction per SIMD at nominal %.2f GHz, 2/4/8 waves per SIMD\n", pr.clockRate * 1e-6);
    all<0>(d, pr.multiProcessorCount, pr.clockRate * 1e-6);
    return 0;
}
