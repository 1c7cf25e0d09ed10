// Does gfx950 apply VGPR-index mode (s_set_gpr_idx_on, M0[7:0]) to (1) arch-VGPR sources/destinations of plain VALU ops and
// (2) the AccVGPR operand of v_accvgpr_read/_write?  Decides how a row of eh[] registers can be walked by a LOOP over
// 8-column blocks instead of fully unrolled code.   hipcc --offload-arch=gfx950 -O3 gpr_idx.hip -o gpr_idx
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void k(uint32_t *out)
{
    const uint32_t lane = threadIdx.x;
    uint32_t r[8];
    // (2) AGPRs a[0..31] = 1000 + 10*k + lane; read a[0..7] under src0 index 8*blk
    asm volatile(
        "v_add_u32 v40, 1000, %8\n"
        ".irp k,0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31\n"
        "v_accvgpr_write_b32 a\\k, v40\n"
        "v_add_u32 v40, 10, v40\n"
        ".endr\n"
        "s_nop 4\n"
        "s_mov_b32 s40, 16\n"
        "s_set_gpr_idx_on s40, 0x1\n"
        "v_accvgpr_read_b32 %0, a0\n v_accvgpr_read_b32 %1, a1\n v_accvgpr_read_b32 %2, a2\n v_accvgpr_read_b32 %3, a3\n"
        "v_accvgpr_read_b32 %4, a4\n v_accvgpr_read_b32 %5, a5\n v_accvgpr_read_b32 %6, a6\n v_accvgpr_read_b32 %7, a7\n"
        "s_set_gpr_idx_off\n"
        : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
        : "v"(lane)
        : "v40", "s40", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17",
          "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31");
    for (int c = 0; c < 8; ++c) out[(0 * 8 + c) * 64 + lane] = r[c];
    // (2b) write under dst index: a[0] written with idx 24 should land in a24; read back a24 directly
    uint32_t w;
    asm volatile(
        "s_mov_b32 s40, 24\n"
        "s_set_gpr_idx_on s40, 0x8\n"
        "v_accvgpr_write_b32 a0, %1\n"
        "s_set_gpr_idx_off\n"
        "s_nop 4\n"
        "v_accvgpr_read_b32 %0, a24\n"
        : "=&v"(w) : "v"(lane + 5000u) : "s40", "a0", "a24");
    out[(1 * 8 + 0) * 64 + lane] = w;
    // (1) arch VGPRs v100..v131 = 2000 + 10*k + lane; v_mov under src0 index 8*blk
    asm volatile(
        "v_add_u32 v40, 2000, %8\n"
        ".irp k,100,101,102,103,104,105,106,107,108,109,110,111,112,113,114,115,116,117,118,119,120,121,122,123,124,125,126,127,128,129,130,131\n"
        "v_mov_b32 v\\k, v40\n"
        "v_add_u32 v40, 10, v40\n"
        ".endr\n"
        "s_mov_b32 s40, 8\n"
        "s_set_gpr_idx_on s40, 0x1\n"
        "v_mov_b32 %0, v100\n v_mov_b32 %1, v101\n v_mov_b32 %2, v102\n v_mov_b32 %3, v103\n"
        "v_mov_b32 %4, v104\n v_mov_b32 %5, v105\n v_mov_b32 %6, v106\n v_mov_b32 %7, v107\n"
        "s_set_gpr_idx_off\n"
        : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
        : "v"(lane)
        : "v40", "s40", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115",
          "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131");
    for (int c = 0; c < 8; ++c) out[(2 * 8 + c) * 64 + lane] = r[c];
    // (1b) a packed op whose src0 is indexed: v_pk_max_u16 d, v100(+idx), v40
    uint32_t pk;
    asm volatile(
        "v_mov_b32 v100, 7\n v_mov_b32 v116, 0x00090003\n v_mov_b32 v40, 0x00050005\n"
        "s_mov_b32 s40, 16\n"
        "s_set_gpr_idx_on s40, 0x1\n"
        "v_pk_max_u16 %0, v100, v40\n"
        "s_set_gpr_idx_off\n"
        : "=&v"(pk) : : "v40", "s40", "v100", "v116");
    out[(3 * 8 + 0) * 64 + lane] = pk;
}

int main()
{
    uint32_t *d, h[4 * 8 * 64];
    (void)hipMalloc(&d, sizeof(h));
    (void)hipMemset(d, 0, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("AccVGPR read, src0 index 16: lane 3 got");
    for (int c = 0; c < 8; ++c) printf(" %u", h[(0 * 8 + c) * 64 + 3]);
    printf("   (indexed: 1163 1173 ..., not indexed: 1003 1013 ...)\n");
    printf("AccVGPR write, dst index 24: a24 reads back %u (indexed: 5003)\n", h[(1 * 8 + 0) * 64 + 3]);
    printf("arch VGPR v_mov, src0 index 8: lane 3 got");
    for (int c = 0; c < 8; ++c) printf(" %u", h[(2 * 8 + c) * 64 + 3]);
    printf("   (indexed: 2083 2093 ..., not indexed: 2003 ...)\n");
    printf("v_pk_max_u16 with indexed src0: 0x%08x (indexed: 0x00090005, not: 0x00050007)\n", h[(3 * 8 + 0) * 64 + 3]);
    return 0;
}
