// Third VALU microbenchmark (round 2): packed-16-bit (VOP3P) ops, 16-bit VOP2 ops, SWAR candidates and short MIXED
// sequences, at 1/2/3/4/8 waves per SIMD — the numbers behind the two-seeds-per-lane lane kernel.
// Prints cycles per wave64 INSTRUCTION per SIMD at the nominal clock.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 2048
#define UNROLL 16
// X(id, instructions per body, asm body).  %0 = x (in/out), %1 = y, %2 = z (all VGPR), s20:s21 scratch SGPRs
#define OPS(X) \
  X(0, 1, "v_pk_max_u16 %0, %0, %1") \
  X(1, 1, "v_pk_sub_u16 %0, %0, %1 clamp") \
  X(2, 1, "v_pk_mad_u16 %0, %0, %1, %2") \
  X(3, 1, "v_pk_lshlrev_b16 %0, 8, %1") \
  X(4, 1, "v_pk_add_u16 %0, %0, %1") \
  X(5, 1, "v_pk_min_u16 %0, %0, %1") \
  X(6, 1, "v_mad_u32_u24 %0, %0, %1, %2") \
  X(7, 1, "v_lshl_or_b32 %0, %0, 8, %1") \
  X(8, 1, "v_or3_b32 %0, %0, %1, %2") \
  X(9, 1, "v_min_u16_e32 %0, %0, %1") \
  X(10, 1, "v_lshlrev_b16_e32 %0, 3, %1") \
  X(11, 1, "v_lshrrev_b16_e32 %0, 3, %1") \
  X(12, 1, "v_mul_lo_u16_e32 %0, %0, %1") \
  X(13, 1, "v_and_b32_e32 %0, 0xff00ff, %1") \
  X(14, 1, "v_lshrrev_b32_e32 %0, 8, %1") \
  X(15, 1, "v_addc_co_u32_e32 %0, vcc, %0, %1, vcc") \
  X(16, 1, "v_cndmask_b32_e32 %0, %0, %1, vcc") \
  X(17, 1, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]") \
  X(18, 1, "v_bfe_u32 %0, %1, 8, 8") \
  X(19, 1, "v_max3_u32 %0, %0, %1, %2") \
  X(20, 1, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0xc8") \
  X(21, 1, "v_max_u32_dpp %0, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf") \
  X(22, 1, "v_pk_max_i16 %0, %0, %1") \
  X(23, 1, "v_pk_sub_i16 %0, %0, %1") \
  X(24, 1, "v_pk_ashrrev_i16 %0, 15, %1") \
  X(25, 1, "v_pk_mul_lo_u16 %0, %0, %1") \
  X(26, 2, "v_and_b32_e32 %0, 0xff00ff, %0\n v_pk_max_u16 %0, %0, %1") \
  X(27, 2, "v_lshrrev_b32_e32 %0, 1, %0\n v_pk_max_u16 %0, %0, %1") \
  X(28, 3, "v_lshrrev_b32_e32 %0, 1, %0\n v_and_b32_e32 %0, 0xff00ff, %0\n v_pk_max_u16 %0, %0, %1") \
  X(29, 2, "v_add_u32_e32 %0, %0, %1\n v_xor_b32_e32 %0, %0, %2") \
  X(30, 2, "v_pk_max_u16 %0, %0, %1\n s_add_u32 s20, s20, 1") \
  X(31, 2, "v_pk_max_u16 %0, %0, %1\n s_lshl_b32 s21, s20, 1") \
  X(32, 2, "v_pk_max_u16 %0, %0, %1\n v_pk_sub_u16 %0, %0, %2 clamp") \
  X(33, 1, "v_pk_max_u16 %0, %0, s20") \
  X(34, 1, "v_sub_u16_e64 %0, %0, %1 clamp") \
  X(35, 1, "v_max_u16_e32 %0, %0, %1") \
  X(36, 1, "v_max_i16_e32 %0, 0x1234, %1") \
  X(37, 1, "v_mov_b32_e32 %0, %1") \
  X(38, 1, "v_sad_u8 %0, %0, %1, %2") \
  X(39, 1, "v_alignbit_b32 %0, %0, %1, 8") \
  X(40, 1, "v_add_u16_e32 %0, %0, %1") \
  X(41, 1, "v_sub_u32_e32 %0, %0, %1") \
  X(42, 1, "v_or_b32_e32 %0, 0x80008, %1") \
  X(43, 1, "v_pk_mad_u16 %0, %0, s20, %2") \
  X(44, 1, "v_pk_add_u16 %0, %0, %1 clamp") \
  X(45, 1, "v_cmp_ne_u32_e64 s[20:21], %0, %1") \
  X(46, 1, "v_cmp_ne_u16_e64 s[20:21], %0, %1") \
  X(47, 1, "v_ffbh_u32_e32 %0, %1") \
  X(48, 1, "v_ffbl_b32_e32 %0, %1") \
  X(49, 1, "v_bcnt_u32_b32 %0, %1, %0")
#define NOPS 50
static const char *names[] = {
#define X(i, n, s) s,
  OPS(X)
#undef X
};
static const int counts[] = {
#define X(i, n, s) n,
  OPS(X)
#undef X
};
template <int KIND>
__global__ __launch_bounds__(256) void k(unsigned *out, unsigned seed)
{
    unsigned a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 7u + i * 13u;
    asm volatile("v_cmp_lt_u32_e32 vcc, %0, %1\n s_mov_b64 s[20:21], vcc" : : "v"(a[0]), "v"(a[1]) : "vcc", "s20", "s21");
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            unsigned &x = a[u & 7], &y = a[(u + 3) & 7], &z = a[(u + 5) & 7];
#define X(i, n, s) if (KIND == i) asm volatile(s : "+v"(x) : "v"(y), "v"(z) : "vcc", "s20", "s21");
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
    printf("%-78s", names[KIND]);
    const int wl[] = {1, 2, 3, 4, 8};
    for (int wi = 0; wi < 5; ++wi) {
        const int wps = wl[wi];
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(cus * wps), dim3(256), 0, 0, d_out, 12345u);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(cus * wps), dim3(256), 0, 0, d_out, 12345u);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf(" w%d: %.2f", wps, ms * 1e-3 * ghz * 1e9 / ((double)ITERS * UNROLL * counts[KIND] * wps));
    }
    printf("\n");
    fflush(stdout);
}
template <int K> static void all(unsigned *d, int cus, double ghz) { run<K>(d, cus, ghz); if constexpr (K + 1 < NOPS) all<K + 1>(d, cus, ghz); }
int main()
{
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    unsigned *d; hipMalloc(&d, (size_t)pr.multiProcessorCount * 8 * 256 * 4);
    printf("cycles per wave64 instruction per SIMD at nominal %.2f GHz, 1/2/3/4/8 waves per SIMD (sequences: per instruction)\n", pr.clockRate * 1e-6);
    all<0>(d, pr.multiProcessorCount, pr.clockRate * 1e-6);
    return 0;
}
