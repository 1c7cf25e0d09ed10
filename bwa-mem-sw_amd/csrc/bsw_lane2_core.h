/*
 * bsw_lane2_core.h — per-lane arithmetic of the TWO-SEEDS-PER-LANE lane kernel (bsw_lane2_kernel.hip).
 *
 * Each lane runs the scalar ksw_extend2 recurrence (sw_pe_array_sw_extend.v:1639-1705, CPU semantics of
 * SURVEY.md §8a, variant H) of two seeds at once: seed A in the low and seed B in the high 16 bits of every
 * register, with packed 16-bit VALU ops (v_pk_max_u16 / v_pk_sub_u16 clamp / v_pk_mad_u16: two values per
 * 4-cycle issue slot — the best per-value rate gfx950 offers for max/add work, profiles/r2/ubench3.txt).
 *   eh[] row: one VGPR per column, each half = {e:8 | h:8} (legal when h0 + qlen*a + b <= 255 and qlen <= 135:
 *   the cell forms H + a + b in the 8 score bits before it subtracts b).
 *   Scores are unsigned with saturating subtraction: max(x - c, 0) is one op.
 *   Left edge: eh[j] == 0 for every j < beg is an invariant (zero-trimming only passes zeros; columns the band
 *   clamp drops are zeroed explicitly) and the match mask is cleared below beg, so cells left of beg compute to
 *   zero and need no masking — only blocks that hold some lane's `end` run the masked ("edge") body.
 *   Next-row range (K8): non-zero bits of the stored eh entries accumulate in 16-column bit masks.
 * This header is compiled twice: by hipcc into the kernel, and by g++ into the CPU model the tests check against
 * the oracle (tests/lane2_model.cpp) — same source, so the arithmetic is verified without a GPU.
 */
#ifndef BSW_LANE2_CORE_H
#define BSW_LANE2_CORE_H

#include <stdint.h>
#include <utility>

#if defined(__HIPCC__)
#define L2_FN __host__ __device__ __forceinline__
#define L2_MFN __host__ __device__ __forceinline__ static
#else
#define L2_FN static inline
#define L2_MFN static inline
#endif

/* profiling build only (-DBSW_L2_STAMP): section time stamps inside the row loop (tools/profile notes in DESIGN.md) */
#if defined(BSW_L2_STAMP) && defined(__HIP_DEVICE_COMPILE__)
#define L2_STAMP(k) bsw_l2_stamp(k)
__device__ void bsw_l2_stamp(int k);
#else
#define L2_STAMP(k) ((void)0)
#endif

namespace bsw {
namespace l2 {

/* ---- packed 16-bit primitives ---- */
#if defined(__HIP_DEVICE_COMPILE__)
L2_FN uint32_t pk_max(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_max_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
L2_FN uint32_t pk_min(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_min_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
L2_FN uint32_t pk_subs(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(d) : "v"(a), "v"(b)); return d; }
L2_FN uint32_t pk_sub(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_sub_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
L2_FN uint32_t pk_mad(uint32_t a, uint32_t b, uint32_t c) { uint32_t d; asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
/* an inline constant feeds only the LOW half of a packed op on gfx9 (its high 16 bits are 0): op_sel_hi:[0,1] makes the
 * high half read the low 16 bits of src0 too */
L2_FN uint32_t pk_shr8(uint32_t a) { uint32_t d; asm("v_pk_lshrrev_b16 %0, 8, %1 op_sel_hi:[0,1]" : "=v"(d) : "v"(a)); return d; }
L2_FN uint32_t pk_shl8(uint32_t a) { uint32_t d; asm("v_pk_lshlrev_b16 %0, 8, %1 op_sel_hi:[0,1]" : "=v"(d) : "v"(a)); return d; }
/* {hi byte of a.lo16, hi byte of b.lo16} per half: (a & 0xff00ff00) | ((b >> 8) & 0x00ff00ff) in one v_perm_b32 */
L2_FN uint32_t pack_hi_bytes(uint32_t a, uint32_t b) { uint32_t d; asm("v_perm_b32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(0x07030501u)); return d; }
L2_FN uint32_t or_vs(uint32_t a, uint32_t sb) { uint32_t d; asm("v_or_b32 %0, %2, %1" : "=v"(d) : "v"(a), "s"(sb)); return d; }
/* 0xffff in every half of a that is non-zero: a * 0xffff, saturated (the inline constant -1 feeds both halves through
 * op_sel_hi) */
L2_FN uint32_t pk_nzmask(uint32_t a) { uint32_t d; asm("v_pk_mad_u16 %0, %1, -1, 0 op_sel_hi:[1,0,0] clamp" : "=v"(d) : "v"(a)); return d; }
/* (a & m) | sc with the mask in a VGPR and the constant in an SGPR (a VOP3 reads one scalar at most on gfx9) */
L2_FN uint32_t and_or_vvs(uint32_t a, uint32_t m, uint32_t sc) { uint32_t d; asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(m), "s"(sc)); return d; }
/* 16-bit half X of lo in bits [15:0], half X of hi in bits [31:16] */
template <int X>
L2_FN uint32_t half_pair(uint32_t lo, uint32_t hi) { uint32_t d; asm("v_perm_b32 %0, %1, %2, %3" : "=v"(d) : "v"(hi), "v"(lo), "s"(X ? 0x07060302u : 0x05040100u)); return d; }
/* byte k of a in bits [7:0], byte k of b in bits [23:16], zeros elsewhere (k = 0..3) */
template <int K>
L2_FN uint32_t byte_pair(uint32_t a, uint32_t b) { uint32_t d; asm("v_perm_b32 %0, %1, %2, %3" : "=v"(d) : "v"(b), "v"(a), "s"(0x0c000c00u | (uint32_t)K | ((uint32_t)(4 + K) << 16))); return d; }
/* variants whose constant operand sits in an SGPR (one scalar per VOP3P instruction on gfx9) */
L2_FN uint32_t pk_subs_vs(uint32_t a, uint32_t sb) { uint32_t d; asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(d) : "v"(a), "s"(sb)); return d; }
L2_FN uint32_t pk_min_vs(uint32_t a, uint32_t sb) { uint32_t d; asm("v_pk_min_u16 %0, %1, %2" : "=v"(d) : "v"(a), "s"(sb)); return d; }
L2_FN uint32_t pk_mad_vsv(uint32_t a, uint32_t sb, uint32_t c) { uint32_t d; asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(sb), "v"(c)); return d; }
L2_FN uint32_t bfi(uint32_t m, uint32_t a, uint32_t b) { uint32_t d; asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "v"(m), "v"(a), "v"(b)); return d; }   /* (m & a) | (~m & b) */
/* (h << 8) | jj with the column constant in an SGPR: no VGPR and no VALU op spent on it */
L2_FN uint32_t key_of(uint32_t h, uint32_t jj) { uint32_t d; asm("v_lshl_or_b32 %0, %1, 8, %2" : "=v"(d) : "v"(h), "s"(jj)); return d; }
/* index of the lowest / highest set bit; 0xffffffff for x == 0 (what the hardware returns) */
L2_FN uint32_t ffbl(uint32_t x) { uint32_t d; asm("v_ffbl_b32 %0, %1" : "=v"(d) : "v"(x)); return d; }
L2_FN uint32_t ffbh(uint32_t x) { uint32_t d; asm("v_ffbh_u32 %0, %1" : "=v"(d) : "v"(x)); return d; }   /* counted from the MSB */
L2_FN int mul24(int a, int b) { int d; asm("v_mul_i32_i24 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }   /* both factors fit 24 bits: one full-rate op, not a 64-bit mad */
L2_FN uint32_t pk_adds_vs(uint32_t a, uint32_t sb) { uint32_t d; asm("v_pk_add_u16 %0, %1, %2 clamp" : "=v"(d) : "v"(a), "s"(sb)); return d; }   /* saturating */
L2_FN uint32_t pk_sub_vs(uint32_t a, uint32_t sb) { uint32_t d; asm("v_pk_sub_u16 %0, %1, %2" : "=v"(d) : "v"(a), "s"(sb)); return d; }          /* wrapping */
L2_FN uint32_t pk_neg(uint32_t a) { uint32_t d; asm("v_pk_sub_u16 %0, 0, %1" : "=v"(d) : "v"(a)); return d; }                                     /* 0 - a per half */
/* byte k of a in bits [7:0], byte k of b in bits [23:16], zeros elsewhere, k = sel & 3 chosen at run time (selector in an SGPR) */
L2_FN uint32_t byte_pair_dyn(uint32_t a, uint32_t b, uint32_t k) { uint32_t d; asm("v_perm_b32 %0, %1, %2, %3" : "=v"(d) : "v"(b), "v"(a), "s"(0x0c000c00u + k * 0x00010001u + 0x00040000u)); return d; }
L2_FN int popc(uint32_t x) { return __builtin_popcount(x); }
L2_FN int clz32(uint32_t x) { return __builtin_clz(x); }
/* (lane2g) the same byte pair with the selector in a VGPR: sel = 0x0c000c00 | kA | (4 + kB) << 16 picks byte kA of a and byte kB of b */
L2_FN uint32_t byte_pair_sel(uint32_t a, uint32_t b, uint32_t sel) { uint32_t d; asm("v_perm_b32 %0, %1, %2, %3" : "=v"(d) : "v"(b), "v"(a), "v"(sel)); return d; }
/* per half: a << (sh & 15), the shift amounts in the halves of sh */
L2_FN uint32_t pk_shlv(uint32_t sh, uint32_t a) { uint32_t d; asm("v_pk_lshlrev_b16 %0, %1, %2" : "=v"(d) : "v"(sh), "v"(a)); return d; }
/* the looped kernel's per-block folds (row maximum, K8 trackers: see lane2l::row_body) as one aligned statement, ordered so
 * that no packed result is read by the instruction behind it */
template <bool GRID = true>
L2_FN void fold8(uint32_t &mk2, uint32_t &Fnz, uint32_t &Lnz, uint32_t mkb, uint32_t nz8, uint32_t J0d, uint32_t J0h, uint32_t ONE2)
{
    uint32_t t0, t1, t2;
#define BSW_FOLD8_BODY "v_pk_min_u16 %[t0], %[nz], %[ONE]\n\t"         \
                       "v_pk_add_u16 %[t1], %[mkb], %[J0d]\n\t"        \
                       "v_pk_mad_u16 %[t0], %[t0], %[J0h], %[nz]\n\t"  \
                       "v_pk_max_u16 %[mk2], %[mk2], %[t1]\n\t"        \
                       "v_pk_sub_u16 %[t2], %[t0], %[ONE]\n\t"         \
                       "v_pk_max_u16 %[L], %[L], %[t0]\n\t"            \
                       "v_pk_min_u16 %[F], %[F], %[t2]"
#define BSW_FOLD8_OPS : [mk2] "+v"(mk2), [F] "+v"(Fnz), [L] "+v"(Lnz), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2) \
                      : [nz] "v"(nz8), [mkb] "v"(mkb), [J0d] "s"(J0d), [J0h] "s"(J0h), [ONE] "s"(ONE2)
    /* (the looped kernel keeps its block loop on the 8-byte fetch grid; the unrolled one does not pad) */
    if (GRID) asm volatile(".p2align 3\n\t" BSW_FOLD8_BODY BSW_FOLD8_OPS);
    else asm volatile(BSW_FOLD8_BODY BSW_FOLD8_OPS);
#undef BSW_FOLD8_BODY
#undef BSW_FOLD8_OPS
}
/* ---- the row scalars of both seeds in one register (pairv below): a few more packed forms, and SDWA forms that read ONE
 * half / byte of a packed register (gfx9 SDWA: VOP1 / VOP2 only) ---- */
L2_FN uint32_t pk_subs_sv(uint32_t sa, uint32_t b) { uint32_t d; asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(d) : "s"(sa), "v"(b)); return d; }   /* scalar - vector, saturating */
L2_FN uint32_t pk_sub_sv(uint32_t sa, uint32_t b) { uint32_t d; asm("v_pk_sub_u16 %0, %1, %2" : "=v"(d) : "s"(sa), "v"(b)); return d; }          /* wrapping */
L2_FN uint32_t pk_adds(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_add_u16 %0, %1, %2 clamp" : "=v"(d) : "v"(a), "v"(b)); return d; }
L2_FN uint32_t pk_mul_sat(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_mad_u16 %0, %1, %2, 0 clamp" : "=v"(d) : "v"(a), "v"(b)); return d; }   /* a * b, 0xffff on overflow */
L2_FN uint32_t pk_mul_sat_vs(uint32_t a, uint32_t sb) { uint32_t d; asm("v_pk_mad_u16 %0, %1, %2, 0 clamp" : "=v"(d) : "v"(a), "s"(sb)); return d; }
L2_FN uint32_t min_halves(uint32_t a) { uint32_t d; asm("v_min_u16_sdwa %0, %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1" : "=v"(d) : "v"(a)); return d; }
L2_FN uint32_t max_halves(uint32_t a) { uint32_t d; asm("v_max_u16_sdwa %0, %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1" : "=v"(d) : "v"(a)); return d; }
/* lowest set bit of byte K of a (0xffffffff: the byte is 0); v_ffbh of the zero-extended byte K (24 + leading zeros inside the byte, 0xffffffff for 0) */
template <int K>
L2_FN uint32_t ffbl_byte(uint32_t a) { uint32_t d; if (K == 0) asm("v_ffbl_b32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0" : "=v"(d) : "v"(a)); else asm("v_ffbl_b32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2" : "=v"(d) : "v"(a)); return d; }
template <int K>
L2_FN uint32_t ffbh_byte(uint32_t a) { uint32_t d; if (K == 0) asm("v_ffbh_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0" : "=v"(d) : "v"(a)); else asm("v_ffbh_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2" : "=v"(d) : "v"(a)); return d; }
/* a + byte K of b (K = 1 or 3: the high byte of a half) */
template <int K>
L2_FN uint32_t add_byte(uint32_t a, uint32_t b) { uint32_t d; if (K == 1) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(d) : "v"(a), "v"(b)); else asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(d) : "v"(a), "v"(b)); return d; }
/* a wave-uniform value the compiler must re-read here: keeps tests on it from being hoisted out of the row loop */
L2_FN uint32_t opaque_s(uint32_t x) { asm volatile("" : "+s"(x)); return x; }
/* the same for a value the compiler may have computed on the vector side although it is wave-uniform */
L2_FN uint32_t opaque_u(uint32_t x) { x = (uint32_t)__builtin_amdgcn_readfirstlane((int)x); asm volatile("" : "+s"(x)); return x; }
#else
L2_FN uint16_t lo16(uint32_t a) { return (uint16_t)a; }
L2_FN uint16_t hi16(uint32_t a) { return (uint16_t)(a >> 16); }
L2_FN uint32_t mk2(uint32_t lo, uint32_t hi) { return (lo & 0xffffu) | (hi << 16); }
L2_FN uint32_t pk_max(uint32_t a, uint32_t b) { return mk2(lo16(a) > lo16(b) ? lo16(a) : lo16(b), hi16(a) > hi16(b) ? hi16(a) : hi16(b)); }
L2_FN uint32_t pk_min(uint32_t a, uint32_t b) { return mk2(lo16(a) < lo16(b) ? lo16(a) : lo16(b), hi16(a) < hi16(b) ? hi16(a) : hi16(b)); }
L2_FN uint32_t pk_subs(uint32_t a, uint32_t b) { return mk2(lo16(a) > lo16(b) ? lo16(a) - lo16(b) : 0, hi16(a) > hi16(b) ? hi16(a) - hi16(b) : 0); }
L2_FN uint32_t pk_sub(uint32_t a, uint32_t b) { return mk2((uint32_t)(lo16(a) - lo16(b)), (uint32_t)(hi16(a) - hi16(b))); }
L2_FN uint32_t pk_mad(uint32_t a, uint32_t b, uint32_t c) { return mk2((uint32_t)lo16(a) * lo16(b) + lo16(c), (uint32_t)hi16(a) * hi16(b) + hi16(c)); }
L2_FN uint32_t pk_shr8(uint32_t a) { return (a >> 8) & 0x00ff00ffu; }
L2_FN uint32_t pk_shl8(uint32_t a) { return (a << 8) & 0xff00ff00u; }
L2_FN uint32_t pack_hi_bytes(uint32_t a, uint32_t b) { return (a & 0xff00ff00u) | ((b >> 8) & 0x00ff00ffu); }
L2_FN uint32_t or_vs(uint32_t a, uint32_t sb) { return a | sb; }
L2_FN uint32_t and_or_vvs(uint32_t a, uint32_t m, uint32_t sc) { return (a & m) | sc; }
L2_FN uint32_t pk_nzmask(uint32_t a) { return ((a & 0xffffu) ? 0xffffu : 0u) | ((a >> 16) ? 0xffff0000u : 0u); }
template <int X>
L2_FN uint32_t half_pair(uint32_t lo, uint32_t hi) { return X ? ((lo >> 16) | (hi & 0xffff0000u)) : ((lo & 0xffffu) | (hi << 16)); }
template <int K>
L2_FN uint32_t byte_pair(uint32_t a, uint32_t b) { return ((a >> (8 * K)) & 0xffu) | (((b >> (8 * K)) & 0xffu) << 16); }
L2_FN uint32_t pk_subs_vs(uint32_t a, uint32_t sb) { return pk_subs(a, sb); }
L2_FN uint32_t pk_min_vs(uint32_t a, uint32_t sb) { return pk_min(a, sb); }
L2_FN uint32_t pk_mad_vsv(uint32_t a, uint32_t sb, uint32_t c) { return pk_mad(a, sb, c); }
L2_FN uint32_t bfi(uint32_t m, uint32_t a, uint32_t b) { return (m & a) | (~m & b); }
L2_FN uint32_t key_of(uint32_t h, uint32_t jj) { return (h << 8) | jj; }
L2_FN uint32_t ffbl(uint32_t x) { return x ? (uint32_t)__builtin_ctz(x) : 0xffffffffu; }
L2_FN uint32_t ffbh(uint32_t x) { return x ? (uint32_t)__builtin_clz(x) : 0xffffffffu; }
L2_FN int mul24(int a, int b) { return a * b; }
L2_FN uint32_t opaque_s(uint32_t x) { return x; }
L2_FN uint32_t opaque_u(uint32_t x) { return x; }
L2_FN uint32_t pk_adds_vs(uint32_t a, uint32_t sb) { const uint32_t lo = lo16(a) + lo16(sb), hi = hi16(a) + hi16(sb); return mk2(lo > 0xffffu ? 0xffffu : lo, hi > 0xffffu ? 0xffffu : hi); }
L2_FN uint32_t pk_sub_vs(uint32_t a, uint32_t sb) { return pk_sub(a, sb); }
L2_FN uint32_t pk_neg(uint32_t a) { return pk_sub(0u, a); }
L2_FN uint32_t byte_pair_dyn(uint32_t a, uint32_t b, uint32_t k) { return ((a >> (8 * k)) & 0xffu) | (((b >> (8 * k)) & 0xffu) << 16); }
L2_FN int popc(uint32_t x) { return __builtin_popcount(x); }
L2_FN int clz32(uint32_t x) { return __builtin_clz(x); }
L2_FN uint32_t byte_pair_sel(uint32_t a, uint32_t b, uint32_t sel) { return ((a >> (8 * (sel & 3u))) & 0xffu) | (((b >> (8 * ((sel >> 16) & 3u))) & 0xffu) << 16); }
L2_FN uint32_t pk_shlv(uint32_t sh, uint32_t a) { return mk2((uint32_t)(uint16_t)(lo16(a) << (lo16(sh) & 15)), (uint32_t)(uint16_t)(hi16(a) << (hi16(sh) & 15))); }
L2_FN uint32_t pk_subs_sv(uint32_t sa, uint32_t b) { return pk_subs(sa, b); }
L2_FN uint32_t pk_sub_sv(uint32_t sa, uint32_t b) { return pk_sub(sa, b); }
L2_FN uint32_t pk_adds(uint32_t a, uint32_t b) { return pk_adds_vs(a, b); }
L2_FN uint32_t pk_mul_sat(uint32_t a, uint32_t b) { const uint32_t lo = (uint32_t)lo16(a) * lo16(b), hi = (uint32_t)hi16(a) * hi16(b); return mk2(lo > 0xffffu ? 0xffffu : lo, hi > 0xffffu ? 0xffffu : hi); }
L2_FN uint32_t pk_mul_sat_vs(uint32_t a, uint32_t sb) { return pk_mul_sat(a, sb); }
L2_FN uint32_t min_halves(uint32_t a) { return lo16(a) < hi16(a) ? lo16(a) : hi16(a); }
L2_FN uint32_t max_halves(uint32_t a) { return lo16(a) > hi16(a) ? lo16(a) : hi16(a); }
template <int K>
L2_FN uint32_t ffbl_byte(uint32_t a) { return ffbl((a >> (8 * K)) & 0xffu); }
template <int K>
L2_FN uint32_t ffbh_byte(uint32_t a) { return ffbh((a >> (8 * K)) & 0xffu); }
template <int K>
L2_FN uint32_t add_byte(uint32_t a, uint32_t b) { return a + ((b >> (8 * K)) & 0xffu); }
template <bool GRID = true>
L2_FN void fold8(uint32_t &mk2, uint32_t &Fnz, uint32_t &Lnz, uint32_t mkb, uint32_t nz8, uint32_t J0d, uint32_t J0h, uint32_t ONE2)
{
    const uint32_t key = pk_mad_vsv(pk_min_vs(nz8, ONE2), J0h, nz8);
    mk2 = pk_max(mk2, mkb + J0d);
    Lnz = pk_max(Lnz, key);
    Fnz = pk_min(Fnz, pk_sub_vs(key, ONE2));
}
#endif

L2_FN uint32_t dup16(int v) { return (uint32_t)(v & 0xffff) * 0x00010001u; }
L2_FN uint32_t pack2(int a, int b) { return ((uint32_t)a & 0xffffu) | ((uint32_t)b << 16); }
L2_FN int imax(int a, int b) { return a > b ? a : b; }
L2_FN int imin(int a, int b) { return a < b ? a : b; }

template <class F, int... I>
L2_FN void sfor_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
L2_FN void sfor(F &&f) { sfor_impl(f, std::make_integer_sequence<int, N>{}); }

/* scoring constants of a launch (wave-uniform) */
struct consts {
    int a, pb, pn;                      /* match score, mismatch penalty (-mat[1]), N penalty (-mat[24]); pb >= pn >= 0 */
    int o_del, e_del, oe_ins, e_ins, zdrop;
    uint32_t OED2s, ED2s, OEI2s, EI2s, ONE2;   /* {oe_del,oe_del} << 8, {e_del,e_del} << 8, the same for insertions, {1,1} */
    uint32_t MC[8];                     /* {a+pb, a+pb} << (8 - c): match bit c of a block byte -> (a + pb) << 8 */
    uint32_t HI2;                       /* 0xff00ff00: the score bytes (kept in a VGPR by the kernel) */
    uint32_t EDp, EIp, ZD2;             /* row tail (K7): {e_del,e_del}, {e_ins,e_ins} unscaled; {zdrop,zdrop}, 0xffff when zdrop is off */
};

/* the packed constants from the scalar ones (o_del + e_del < 256, o_ins + e_ins < 256 and a + pb < 256: lane2_params_ok) */
L2_FN void fill_packed_consts(consts &k)
{
    k.OED2s = dup16((k.o_del + k.e_del) << 8);
    k.ED2s = dup16(k.e_del << 8);
    k.OEI2s = dup16(k.oe_ins << 8);
    k.EI2s = dup16(k.e_ins << 8);
    k.ONE2 = 0x00010001u;
    for (int c = 0; c < 8; ++c) k.MC[c] = dup16((k.a + k.pb) << (8 - c));
    k.HI2 = 0xff00ff00u;
    k.EDp = dup16(k.e_del);
    k.EIp = dup16(k.e_ins);
    k.ZD2 = k.zdrop > 0 ? dup16(k.zdrop < 0xffff ? k.zdrop : 0xffff) : 0xffffffffu;
}

#if defined(__HIP_DEVICE_COMPILE__) && defined(BSW_L2_ASM_BODY)
/* the 8-column block bodies as one list-scheduled asm statement each (tools/gen_lane2_body.py) */
#include "bsw_lane2_body_asm.inc"
#endif

/* The scalars of the TWO ksw_extend2 calls of a lane (K1/K9), packed like everything else: seed A in the low, seed B in the
 * high 16 bits.  Round 5: the row head (K3) and tail (K7, K8) used to run per seed in int32 — ~230 VALU instructions per
 * row behind the cell blocks, a quarter of a short row — and now run once for both seeds in packed 16-bit ops: compare =
 * saturating subtract + "non-zero -> 0xffff" (v_pk_mad_u16 x, -1, 0 clamp), select = one bit-select.  Everything fits 16
 * bits: columns < 256, rows < 65535 (BSW_MAX_TLEN), scores < 256 in these classes; the values that start at -1 are kept + 1. */
struct pairv {
    uint32_t BEG, END;                  /* [beg, end) of the current row */
    uint32_t QLEN, TLEN, W, W1, H0;     /* constants of the call; W = min(w, 0xffff), W1 = min(w + 1, 0xffff) */
    uint32_t MX, MAXI1, MAXJ1, MAXIE1, GS1, MOFF;   /* max, max_i + 1, max_j + 1, max_ie + 1, gscore + 1, max_off */
    uint32_t ALIVE;                     /* 0xffff in a half whose band try still has rows */
    uint32_t CELLS16;                   /* cells of the rows since the last flush (< 64 rows x 232 columns) */
    unsigned cells[2];
};

struct u4 {                             /* four consecutive 32-bit words (one 128-bit LDS read) */
    uint32_t v[4];
};

struct rowp {                           /* per-lane values of the current row, both seeds packed */
    uint32_t ACT, BITE;                 /* 0xffff per half: the seed has a row i / the band clamp moved its beg */
    uint32_t ZLO, ZHI;                  /* the columns [zlo, zhi) the clamp dropped */
};

L2_FN int half_of(uint32_t v, int x) { return (int)((v >> (16 * x)) & 0xffffu); }
L2_FN void set_half(uint32_t &v, int x, uint32_t val) { v = x ? ((v & 0x0000ffffu) | (val << 16)) : ((v & 0xffff0000u) | (val & 0xffffu)); }

/* one side of one seed into half x (K1: sw_pe_array_sw_extend.v:889,919,1009,929: max = h0, max_i = max_j = -1, max_off = 0) */
L2_FN void init_pair(pairv &p, int x, int qlen, int tlen, int h0, int w)
{
    if (x == 0) { p.BEG = p.END = p.QLEN = p.TLEN = p.W = p.W1 = p.H0 = p.MX = p.MAXI1 = p.MAXJ1 = p.MAXIE1 = p.GS1 = p.MOFF = p.ALIVE = p.CELLS16 = 0; }
    set_half(p.END, x, (uint32_t)qlen); set_half(p.QLEN, x, (uint32_t)qlen); set_half(p.TLEN, x, (uint32_t)tlen);
    set_half(p.W, x, (uint32_t)imin(w, 0xffff)); set_half(p.W1, x, (uint32_t)imin(w, 0xfffe) + 1u);
    set_half(p.H0, x, (uint32_t)h0); set_half(p.MX, x, (uint32_t)h0);
    set_half(p.ALIVE, x, tlen > 0 ? 0xffffu : 0u);
    p.cells[x] = 0;
}

struct ext_out { int mx, max_i, max_j, max_ie, gscore, max_off; unsigned cells; };
/* K9: the outputs of half x */
L2_FN ext_out pair_result(const pairv &p, int x)
{
    ext_out e;
    e.mx = half_of(p.MX, x); e.max_i = half_of(p.MAXI1, x) - 1; e.max_j = half_of(p.MAXJ1, x) - 1;
    e.max_ie = half_of(p.MAXIE1, x) - 1; e.gscore = half_of(p.GS1, x) - 1; e.max_off = half_of(p.MOFF, x);
    e.cells = p.cells[x] + (unsigned)half_of(p.CELLS16, x);
    return e;
}

/* K3 band clamp for both seeds (:1803,1894-1897,1842,1898): beg = max(beg, i - w), end = min(end, i + w + 1, qlen).  A clamp
 * that moves beg may drop non-zero eh entries: the columns [zlo, zhi) it drops are zeroed by zero_dropped() to keep the
 * "eh[j] == 0 below beg" invariant.  I2 = {i, i} (wave-uniform). */
L2_FN void row_begin2(pairv &p, const int i, rowp &r)
{
    const uint32_t I2 = dup16(i);
    if (__builtin_expect((i & 63) == 0, 0)) {        /* (rare) the 16-bit cell counters into the 32-bit ones */
        p.cells[0] += p.CELLS16 & 0xffffu; p.cells[1] += p.CELLS16 >> 16; p.CELLS16 = 0;
    }
    const uint32_t act = p.ALIVE & pk_nzmask(pk_subs_vs(p.TLEN, I2));          /* alive and i < tlen */
    const uint32_t nb = pk_max(p.BEG, pk_subs_sv(I2, p.W));
    const uint32_t ne = pk_min(pk_min(p.END, pk_adds_vs(p.W1, I2)), p.QLEN);
    r.ACT = act;
    r.BITE = act & pk_nzmask(pk_subs(nb, p.BEG));
    r.ZLO = p.BEG; r.ZHI = nb;
    p.BEG = (act & nb) | (~act & p.BEG);
    p.END = (act & ne) | (~act & p.END);
    p.CELLS16 += pk_subs(p.END, p.BEG) & act;        /* (no carry between the halves: flushed every 64 rows) */
}

/* K7 + K8 for both seeds, after the row's cells: h1 = H(i, end - 1) scaled (high byte of each half), mk2 = (row max << 8) |
 * its column per half (0: no positive cell), Fnz / Lnz = the packed first / last non-zero trackers of the row's blocks:
 *   Lnz half = (j0 << 8) | bits of the HIGHEST block with a non-zero stored eh entry (j0 = 8 b its first column), 0: none
 *   Fnz half = the same key of the LOWEST such block, minus 1; 0xffff: none.
 * (:1829-1833 gscore / max_ie, ties -> later i; :1959,1810,1845 new maximum; zdrop; :1942 m == 0; K8: CPU semantics of
 * SURVEY.md §8a, quirk Q5 avoided.) */
template <bool SYM>
L2_FN void row_tail2(pairv &p, const consts &k, const int i, const uint32_t act, const uint32_t h1, const uint32_t mk2,
                     const uint32_t Fnz, const uint32_t Lnz)
{
    const uint32_t I2 = dup16(i), IP1 = dup16(i + 1);
    const uint32_t H1P = pk_shr8(h1) + k.ONE2;                           /* eh[end].h + 1 (<= 256: no carry between the halves) */
    const uint32_t M = pk_shr8(mk2), MJ = mk2 & 0x00ff00ffu, MJ1 = MJ + k.ONE2;
    /* K7: the row reached the end of the query */
    const uint32_t atq = act & ~pk_nzmask(pk_max(p.BEG, p.END) ^ p.QLEN);
    const uint32_t sie = atq & ~pk_nzmask(pk_subs(p.GS1, H1P));          /* ... and h1 >= gscore: ties -> later i */
    p.MAXIE1 = (sie & IP1) | (~sie & p.MAXIE1);
    const uint32_t gsm = pk_max(p.GS1, H1P);
    p.GS1 = (atq & gsm) | (~atq & p.GS1);
    /* K7: a new maximum; zdrop against the OLD one */
    const uint32_t gt = act & pk_nzmask(pk_subs(M, p.MX));
    const uint32_t off = pk_subs_sv(I2, MJ) | pk_subs_vs(MJ, I2);        /* |mj - i| */
    const uint32_t A = pk_adds(pk_sub_sv(IP1, p.MAXI1), p.MAXJ1);        /* (i - max_i) + (max_j + 1); saturated = far beyond any zdrop */
    const uint32_t dpos = pk_subs(A, MJ1), dneg = pk_subs(MJ1, A);       /* dd = (i - max_i) - (mj - max_j): its positive / negative part */
    const uint32_t ad = dpos | dneg;
    uint32_t prod;
    if (SYM) prod = pk_mul_sat_vs(ad, k.EDp);
    else { const uint32_t pos = pk_nzmask(dpos); prod = pk_mul_sat(ad, (pos & k.EDp) | (~pos & k.EIp)); }   /* rows ahead: a deletion's penalty */
    const uint32_t lhs = pk_subs_vs(pk_subs(p.MX, M), k.ZD2);            /* mx - m - zdrop, 0 when not positive (or zdrop off) */
    const uint32_t zstop = pk_nzmask(pk_subs(lhs, prod)) & ~gt;
    const uint32_t stop = ~pk_nzmask(M) | zstop;                         /* (:1942) */
    const uint32_t mo = pk_max(p.MOFF, off);
    p.MOFF = (gt & mo) | (~gt & p.MOFF);
    p.MAXI1 = (gt & IP1) | (~gt & p.MAXI1);
    p.MAXJ1 = (gt & MJ1) | (~gt & p.MAXJ1);
    p.MX = (gt & M) | (~gt & p.MX);
    /* K8 next-row range: first / last non-zero eh entry in [beg, end] */
    const uint32_t nF = ~Fnz;                                            /* low byte of a half: ~(bits - 1): its trailing zeros = ctz(bits) */
    const uint32_t f0 = add_byte<1>(ffbl_byte<0>(nF), Fnz), f1 = add_byte<3>(ffbl_byte<2>(nF), Fnz);   /* j0 + ctz(bits); 254 when none */
    const uint32_t nbeg = pk_min(f0 | (f1 << 16), p.END);                /* (f0, l0 < 2^16: no mask needed) */
    const uint32_t l0 = add_byte<1>(32u - ffbh_byte<0>(Lnz), Lnz), l1 = add_byte<3>(32u - ffbh_byte<2>(Lnz), Lnz);   /* last + 1 */
    const uint32_t LN1 = (l0 | (l1 << 16)) & pk_nzmask(Lnz);             /* 0 when none */
    const uint32_t nend = pk_min(pk_max(LN1, nbeg) + k.ONE2, p.QLEN);    /* min(last + 2, qlen), last = nbeg - 1 when nothing is left */
    p.BEG = (act & nbeg) | (~act & p.BEG);
    p.END = (act & nend) | (~act & p.END);
    p.ALIVE = (act & ~stop) | (~act & p.ALIVE);
}

struct uni {                            /* wave-uniform values of the current row */
    int jlo, jhi, jem;                  /* min beg, max end, min end over the active seeds of the wave */
    bool anybite;
    int zl, zh;                         /* union of the column ranges the band clamp dropped this row */
    uint32_t nblk;                      /* bit b: some query of the wave has an N in columns [8b, 8b+8) */
};

/* VM: variant M (bwa >= 0.7.9: M = H(i-1,j-1) ? H(i-1,j-1) + s : 0, gaps open from M) instead of variant H (the RTL's,
 * sw_pe_array_sw_extend.v:1797-1798,1863,1866: gaps open from h).  SYM: o_del == o_ins and e_del == e_ins (one shared
 * gap-open term per cell); the RTL's datapath takes the four penalties separately (sw_pe_array_proc_element.v:816-819). */
template <int QB, bool VM = false, bool SYM = true>
struct lane2 {
    static constexpr int QMAX = QB * 8;
    static constexpr int NW = (QMAX + 31) / 32;     /* 32-column match-mask words per seed */
    static constexpr int NC = (QMAX + 15) / 16;     /* 16-column chunks (both seeds per register) */
    static constexpr int NG = (QMAX + 63) / 64;     /* 64-column groups of the row-max key */

    struct state {
        uint32_t Pr[QMAX];              /* eh[] row: half = {e:8 | h:8} */
        pairv p;                        /* the scalars of both seeds, packed */
    };

    /* K2 first row, closed form: eh[0]=h0, eh[j]=max(h0-oe_ins-(j-1)e_ins,0), e=0 (sw_pe_array_sw_extend.v:1979,1957,1974) */
    L2_MFN void init_row(state &S, const consts &k)
    {
        const int h00 = half_of(S.p.H0, 0), h01 = half_of(S.p.H0, 1);
        sfor<QMAX>([&](auto ci) {
            constexpr int j = decltype(ci)::value;
            const int v0 = j == 0 ? h00 : imax(h00 - k.oe_ins - (j - 1) * k.e_ins, 0);
            const int v1 = j == 0 ? h01 : imax(h01 - k.oe_ins - (j - 1) * k.e_ins, 0);
            S.Pr[j] = pack2(v0, v1);
        });
    }

    L2_MFN void zero_dropped(state &S, const rowp &r, const uni &u)
    {
        const bool b0 = (r.BITE & 0xffffu) != 0, b1 = (r.BITE >> 16) != 0;
        const int zlo0 = half_of(r.ZLO, 0), zlo1 = half_of(r.ZLO, 1), zhi0 = half_of(r.ZHI, 0), zhi1 = half_of(r.ZHI, 1);
        sfor<QB>([&](auto bi) {
            constexpr int j0 = decltype(bi)::value * 8;
            if (j0 + 8 <= u.zl || j0 >= u.zh) return;
            sfor<8>([&](auto ci) {
                constexpr int J = j0 + decltype(ci)::value;
                uint32_t keep = 0xffffffffu;
                if (b0 && J >= zlo0 && J < zhi0) keep &= 0xffff0000u;
                if (b1 && J >= zlo1 && J < zhi1) keep &= 0x0000ffffu;
                S.Pr[J] &= keep;
            });
        });
    }

    /* One DP cell of column J for both seeds (K5: :1797-1798,1809,1866,1863,1776).
     * Scores live in the HIGH byte of each 16-bit half inside the row ("scaled": value << 8, low byte 0); saturating
     * subtract and max are scale-invariant, the match bit needs no normalising shift (bit c of the block's match
     * byte times (a+b) << (8-c) is (a+b) << 8 whatever c is), the row-max key is `h | column` and the stored pair
     * {e', H(i,j-1)} is one byte permute.  The LOW byte of a scaled value is don't-care: 16-bit max and saturating
     * subtract are exact on the high byte whatever the low bytes hold (a tie in the high byte is a tie in the
     * score), so eh[j] = {e:8 | h:8} itself serves as "e" without an extraction; only the key and the stored pair
     * need clean bytes, and both take them by byte selection.
     * EDGE: the block holds some seed's `end` -> writes, the row max and the non-zero bits are masked per half to
     *       J < end (cells) / J <= end (the eh[end] = {h1, 0} store, :1775).  NQ: some query has an N in this block.
     * State flows through h1, f (scaled), mk, nz. */
    template <int J, bool EDGE, bool NQ>
    L2_MFN void cell(uint32_t &P, const uint32_t Wc8, const uint32_t WNc, const uint32_t Bv2s, const uint32_t D2s,
                           const consts &k, const uint32_t END2, uint32_t &mi_prev, uint32_t &h1, uint32_t &f, uint32_t &mk, uint32_t &nz)
    {
        /* key and non-zero bit are relative to the 8-column block: 16 distinct SGPR constants in the whole kernel
         * (column-absolute ones would be ~80, all hoisted out of the row loop, and spill) */
        constexpr int C = J & 7;
        constexpr uint32_t JJ = (uint32_t)C * 0x00010001u;
        constexpr uint32_t BIT = (uint32_t)(1u << C) * 0x00010001u;
        const uint32_t t = Wc8 & BIT;                        /* 2^C where q_j == t_i */
        const uint32_t hd = pk_shl8(P);                      /* eh[j].h = H(i-1,j-1), scaled */
        const uint32_t e = P;                                /* eh[j].e in the high byte (the low byte is don't-care) */
        uint32_t X = pk_mad_vsv(t, k.MC[C], hd);             /* hd + (match ? a + pb : 0) */
        if (NQ) {
            uint32_t n = (J & 15) ? (WNc >> (J & 15)) : WNc;
            n &= 0x00010001u;
            X = pk_mad(n, D2s, X);                           /* a query N scores -pn whatever the target base is */
        }
        uint32_t M = pk_subs(X, Bv2s);                       /* max(hd + s, 0): e and f are >= 0 anyway */
        if (VM) M &= pk_nzmask(hd);                          /* variant M: a zero H(i-1,j-1) stays zero (only a match could lift it) */
        uint32_t h = pk_max(pk_max(M, e), f);                /* (:1798,1809) */
        const uint32_t g = VM ? M : h;                       /* what a gap opens from: h in variant H (:1863,1866), M in variant M */
        const uint32_t tD = pk_subs_vs(g, k.OED2s);
        const uint32_t tI = SYM ? tD : pk_subs_vs(g, k.OEI2s);
        uint32_t en = pk_max(pk_subs_vs(e, k.ED2s), tD);     /* (:1866,1770-1771) */
        f = pk_max(pk_subs_vs(f, SYM ? k.ED2s : k.EI2s), tI);   /* (:1863,1780-1781) */
        if (!EDGE) {
            const uint32_t key = and_or_vvs(h, k.HI2, JJ);   /* row max of this block, ties -> later j */
            mk = C ? pk_max(mk, key) : key;
            const uint32_t np = pack_hi_bytes(en, h1);       /* eh[j] = {e', H(i,j-1)} (:1776) */
            const uint32_t nb = pk_min_vs(np, k.ONE2);
            nz = C ? pk_mad_vsv(nb, BIT, nz) : nb;
            P = np;
            h1 = h;
        } else {
            const uint32_t d = pk_subs_vs(END2, dup16(C));   /* END2 is relative to the block here: non-zero iff J < end */
            const uint32_t mi = pk_nzmask(d);                /* 0xffff where J < end */
            const uint32_t mw = mi_prev;                     /* 0xffff where J <= end: J - 1 < end, the previous column's mask */
            mi_prev = mi;
            const uint32_t key = and_or_vvs(h & mi, k.HI2, JJ);
            mk = C ? pk_max(mk, key) : key;
            en &= mi;
            const uint32_t np = pack_hi_bytes(en, h1) & mw;
            const uint32_t nb = pk_min_vs(np, k.ONE2);
            nz = C ? pk_mad_vsv(nb, BIT, nz) : nb;
            P = bfi(mw, np, P);
            h1 = bfi(mi, h, h1);
        }
    }

    /* The eight cells of one block.  WN: the N bits of the block's columns in bits 0..7 of each half (NQ only); END /
     * mi_in: END2 relative to the block and the mask of column j0 - 1 (EDGE only).  On the GPU, with BSW_L2_ASM_BODY, one
     * hand-scheduled asm statement (bsw_lane2_body_asm.inc) instead of eight cell() calls: same instructions, ordered for
     * instruction-level parallelism (what the one-wave-per-SIMD kernel needs). */
    template <bool EDGE, bool NQ>
    L2_MFN void block8(uint32_t (&T)[8], const uint32_t Wc, const uint32_t WN, const uint32_t Bv2s, const uint32_t D2s, const consts &k,
                       const uint32_t END, const uint32_t mi_in, uint32_t &h1, uint32_t &f, uint32_t &mk, uint32_t &nz)
    {
#if defined(__HIP_DEVICE_COMPILE__) && defined(BSW_L2_ASM_BODY)
        if constexpr (EDGE && NQ) block8_asm<true, true, VM, SYM>::run(T, Wc, WN, D2s, Bv2s, k, END, mi_in, h1, f, mk, nz);
        else if constexpr (EDGE) block8_asm<true, false, VM, SYM>::run(T, Wc, Bv2s, k, END, mi_in, h1, f, mk, nz);
        else if constexpr (NQ) block8_asm<false, true, VM, SYM>::run(T, Wc, WN, D2s, Bv2s, k, h1, f, mk, nz);
        else block8_asm<false, false, VM, SYM>::run(T, Wc, Bv2s, k, h1, f, mk, nz);
#else
        uint32_t mi_prev = mi_in;
        sfor<8>([&](auto ci) { cell<decltype(ci)::value, EDGE, NQ>(T[decltype(ci)::value], Wc, WN, Bv2s, D2s, k, END, mi_prev, h1, f, mk, nz); });
#endif
    }

    /* The ragged blocks of a row, column by column with a wave-uniform guard (no query N in the block):
     *   !EDGE: the row's FIRST block — columns below `guard` = jlo & 7 lie left of every active seed's beg; they would
     *          compute zeros from zeros (h1 = f = 0 there: a seed with beg > 0 starts its row from 0) and are skipped;
     *    EDGE: the row's LAST block — columns above `guard` = jhi & 7 lie right of every active seed's `end` column;
     *          every mask is off there (nothing stored, nothing counted) and they are skipped.
     * On the GPU one asm statement with scalar branches between the columns (bsw_lane2_body_asm.inc). */
    template <bool EDGE>
    L2_MFN void block8_seq(uint32_t (&T)[8], const uint32_t Wc, const uint32_t Bv2s, const consts &k, const uint32_t END, const uint32_t mi_in,
                           const int guard, uint32_t &h1, uint32_t &f, uint32_t &mk, uint32_t &nz)
    {
#if defined(__HIP_DEVICE_COMPILE__) && defined(BSW_L2_ASM_BODY)
        if constexpr (EDGE) block8_seq_asm<true, VM, SYM>::run(T, Wc, Bv2s, k, END, mi_in, guard, h1, f, mk, nz);
        else block8_seq_asm<false, VM, SYM>::run(T, Wc, Bv2s, k, guard, h1, f, mk, nz);
#else
        uint32_t mi_prev = mi_in;
        mk = 0; nz = 0;
        sfor<8>([&](auto ci) {
            constexpr int c = decltype(ci)::value;
            if (EDGE ? c > guard : c < guard) return;
            /* (column 0 of cell() assigns mk / nz instead of folding: fold by hand when it was not the first one run) */
            uint32_t mkc = 0, nzc = 0;
            cell<c, EDGE, false>(T[c], Wc, 0u, Bv2s, 0u, k, END, mi_prev, h1, f, c ? mk : mkc, c ? nz : nzc);
            if (c == 0) { mk = mkc; nz = nzc; }
        });
#endif
    }

    /* match-mask words of one seed for target base tb: bit j = (q_j == t_i), neither an N, j >= beg.
     * mw(x, b, rm) loads the NW words precomputed for base b (0..3) of seed x's query: the kernel keeps the four
     * per-base masks in LDS, so a row costs two LDS reads per seed and no plane arithmetic. */
    static constexpr int KEEP_NONE = 32 * NW;         /* index of the all-zero entry of the keep table */

    /* word wd of the keep-mask table entry b: bits of the columns >= b (entry KEEP_NONE: nothing) */
    L2_MFN uint32_t keep_word(int b, int wd)
    {
        return b <= 32 * wd ? 0xffffffffu : (b >= 32 * wd + 32 ? 0u : 0xffffffffu << (b - 32 * wd));
    }

    /* kp(b, kw) loads the NW words of keep-table entry b (the kernel keeps the table in LDS, one per workgroup): the
     * columns below beg lose their match bits with one AND per word; a row against a target N (no match anywhere)
     * takes the all-zero entry. */
    template <class MW, class KP>
    L2_MFN void match_words(const MW &mw, const KP &kp, int x, int tb, int beg, uint32_t (&rm)[NW])
    {
        mw(x, tb & 3, rm);
        uint32_t kw[NW];
        kp(tb < 4 ? beg : KEEP_NONE, kw);
        sfor<NW>([&](auto wi) {
            constexpr int wd = decltype(wi)::value;
            rm[wd] &= kw[wd];
        });
    }

    /* the four per-base match words of one 32-column query word from its bit-planes (code bit 0, bit 1, N) */
    L2_MFN uint32_t base_match(uint32_t p0, uint32_t p1, uint32_t p2, int b)
    {
        return ((b & 1) ? p0 : ~p0) & ((b & 2) ? p1 : ~p1) & ~p2;
    }

    /* One DP row for both seeds, after the wave-uniform values are known.  tb[x] = target base of seed x (0..4).
     * qp(x, b, rm): per-base match words of seed x, kp(b, kw): keep-table entry (see match_words); wn(c): N planes of both seeds interleaved per
     * 16-column chunk (low half seed A). */
    template <class QP, class KP, class WN>
    L2_MFN void row_body(state &S, const consts &k, const int i, const rowp &r, const uni &u, const int (&tb)[2],
                               const QP &qp, const KP &kp, const WN &wn)
    {
        pairv &p = S.p;
        if (__builtin_expect(u.anybite, 0)) zero_dropped(S, r, u);   /* (rare; 136 masked moves kept out of the hot path's layout) */
        uint32_t rmA[NW], rmB[NW];
        match_words(qp, kp, 0, tb[0], (int)(p.BEG & 0xffffu), rmA);
        match_words(qp, kp, 1, tb[1], (int)(p.BEG >> 16), rmB);
        /* a row against a target N scores -pn everywhere (mat[4][.], :1915-1940) */
        /* (the match multiplier stays a + pb: against a target N no match bit is set) */
        const int pbA = tb[0] < 4 ? k.pb : k.pn, pbB = tb[1] < 4 ? k.pb : k.pn;
        const uint32_t Bv2 = pack2(pbA, pbB) << 8, D2 = pack2(pbA - k.pn, pbB - k.pn) << 8;
        const uint32_t END2 = p.END;
        /* K4 column 0 (:1795-1796,1835), CPU semantics: only when beg == 0; h0 - (o_del + e_del (i + 1)), both seeds at once */
        const uint32_t h1c = pk_subs_vs(p.H0, dup16(imin(k.o_del + k.e_del * (i + 1), 0xffff)));
        uint32_t h1 = pk_shl8(h1c & ~pk_nzmask(p.BEG)), f = 0;      /* scaled, like every score inside the row */
        L2_STAMP(2);
#if defined(BSW_L2_PRIO) && defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_s_setprio(0);                      /* (experiment) the block phase yields to a wave in its serial row head / tail */
#endif
        uint32_t mkg[NG];
        sfor<NG>([&](auto gi) { mkg[decltype(gi)::value] = 0; });
        uint32_t Fnz = 0xffffffffu, Lnz = 0;                /* K8: packed first / last non-zero trackers (row_tail2) */

        const int blo = u.jlo >> 3, bhi = u.jhi >> 3, bem = u.jem >> 3;      /* block granularity of the dispatch */
        const uint32_t nblk = opaque_s(u.nblk);
        /* Which body each block takes, as one bit mask per body, set up ONCE per row: per block the dispatch is then a bit
         * test and a branch per candidate body.  (Written as comparisons of b with blo / bhi / bem inside the block loop, the
         * compiler rebuilt ~25 scalar instructions and five branches around every block — a sixth of a wave's time in
         * the blocks, during which the SIMD's other wave issues at the lone-wave rate.) */
#ifndef BSW_L2_RAGGED_TOP
#define BSW_L2_RAGGED_TOP 10        /* with the bit-mask dispatch: 4 / 8 / 10 / 12 / 17 -> headline 2681 / 2663 / 2675 / 2650 / 2619, 72-column bin 2127 / 2350 / 2345 / 2318 / 2317, mixed bins 2305 / 2330 / 2340 / 2319 / 2352 */
#endif
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BSW_L2_RAGGED_UNROLLED)
#ifndef BSW_L2_RAGGED_FIRST
#define BSW_L2_RAGGED_FIRST 0
#endif
        constexpr uint32_t RAG_F = (1u << BSW_L2_RAGGED_FIRST) - 1u, RAG_L = ((1u << QB) - 1u) & ~((1u << (QB > BSW_L2_RAGGED_TOP ? QB - BSW_L2_RAGGED_TOP : 0)) - 1u);   /* only the last block, only at the top of the class */
#else
        constexpr uint32_t RAG_F = (1u << QB) - 1u, RAG_L = (1u << QB) - 1u;
#endif
        const uint32_t ALLB = (1u << QB) - 1u;
        const uint32_t m_run = ALLB & ~((1u << imin(blo, 31)) - 1u) & ((2u << imin(bhi, 30)) - 1u);        /* blo <= b <= bhi */
        const uint32_t m_lt_em = (1u << imin(imax(bem, 0), 31)) - 1u;                                       /* b < bem: dense */
        const uint32_t m_seqf = ((u.jlo & 7) != 0 ? (1u << imin(blo, 31)) : 0u) & RAG_F & m_lt_em & ~nblk & m_run;
        const uint32_t m_seql = ((u.jhi & 7) != 7 ? (1u << imin(bhi, 31)) : 0u) & RAG_L & ~m_lt_em & ~nblk & m_run;
        const uint32_t m_dense = opaque_u(m_run & m_lt_em & ~nblk & ~m_seqf);
        const uint32_t m_edge = opaque_u(m_run & ~m_lt_em & ~nblk & ~m_seql);
        const uint32_t m_dnq = opaque_u(m_run & m_lt_em & nblk);
        const uint32_t m_enq = opaque_u(m_run & ~m_lt_em & nblk);
        const uint32_t m_sf = opaque_u(m_seqf), m_sl = opaque_u(m_seql), m_any = opaque_u(m_run), m_cold = opaque_u(m_run & nblk);
        sfor<QB>([&](auto bi) {
            constexpr int b = decltype(bi)::value, j0 = b * 8, g = j0 >> 6, c = j0 >> 4, wd = j0 >> 5;
            if (!((m_any >> b) & 1u)) return;                 /* j0 + 8 <= jlo or j0 > jhi */
            /* both seeds' match bits of this 16-column chunk: low half seed A, high half seed B */
            const uint32_t Wc = byte_pair<b & 3>(rmA[wd], rmB[wd]);   /* this block's 8 match bits of seed A / B in bits 0..7 of the low / high half */
            uint32_t mkb = 0, nz8 = 0;                        /* this block's row-max key and non-zero bits */
            /* nested scalar branches (a combined condition would be materialised as lane masks for all 17 blocks
             * before the loop and spill) */
            /* four mutually exclusive bodies as four consecutive `if`s, not an if/else tree: a body that is simply run or
             * skipped updates eh[j] in place, an if/else joins differently allocated versions with a v_mov per column */
            const uint32_t dummy = 0;
            /* the block's eight columns go through a local array (values, not memory: the bodies take them by reference) */
            const auto run8 = [&](auto edge, auto nqv, const uint32_t WNc, const uint32_t ENDx, const uint32_t mi_in) {
                uint32_t T[8];
                sfor<8>([&](auto ci) { T[decltype(ci)::value] = S.Pr[j0 + decltype(ci)::value]; });
                block8<decltype(edge)::value, decltype(nqv)::value>(T, Wc, WNc, Bv2, D2, k, ENDx, mi_in, h1, f, mkb, nz8);
                sfor<8>([&](auto ci) { S.Pr[j0 + decltype(ci)::value] = T[decltype(ci)::value]; });
            };
            using no_t = std::integral_constant<bool, false>;
            using yes_t = std::integral_constant<bool, true>;
            /* the ragged first / last block of the row column by column (block8_seq).  Exact either way (the CPU model runs
             * both).  The looped kernel uses both (250 bp: +4.7 %).  Unrolled, two more bodies for EVERY block grow the code
             * from 213 to 334 KB and the 150 bp headline drops from 2 360 to 2 215 GCUPS (instruction cache); the last-block
             * body for the top BSW_L2_RAGGED_TOP blocks of the class alone — where the queries of the class end — lifts it
             * to 2 478 (sweep 1 / 2 / 3 / 4 / 6 / 17 blocks: 2 384 / 2 461 / 2 478 / 2 478 / 2 377 / 2 313,
             * profiles/r3/ragged_top_sweep.txt) */
            if ((m_dense >> b) & 1u) run8(no_t{}, no_t{}, 0u, END2, dummy);
            if (RAG_F != 0u && ((m_sf >> b) & 1u)) {
                uint32_t T[8];
                sfor<8>([&](auto ci) { T[decltype(ci)::value] = S.Pr[j0 + decltype(ci)::value]; });
                block8_seq<false>(T, Wc, Bv2, k, END2, dummy, u.jlo & 7, h1, f, mkb, nz8);
                sfor<8>([&](auto ci) { S.Pr[j0 + decltype(ci)::value] = T[decltype(ci)::value]; });
            }
            if (((RAG_L >> b) & 1u) != 0u && ((m_sl >> b) & 1u)) {
                const uint32_t d0 = pk_subs_vs(END2 + 0x00010001u, dup16(j0));
                const uint32_t ENDr = pk_subs_vs(END2, dup16(j0));
                uint32_t T[8];
                sfor<8>([&](auto ci) { T[decltype(ci)::value] = S.Pr[j0 + decltype(ci)::value]; });
                block8_seq<true>(T, Wc, Bv2, k, ENDr, pk_nzmask(d0), u.jhi & 7, h1, f, mkb, nz8);
                sfor<8>([&](auto ci) { S.Pr[j0 + decltype(ci)::value] = T[decltype(ci)::value]; });
            }
            /* (query-N bodies: unlikely, the bins keep the queries with an N apart — one test for both, cold code out of line) */
            if (__builtin_expect((m_cold >> b) & 1u, 0)) {
            if ((m_dnq >> b) & 1u) {
                const uint32_t WNr = wn(c);
                run8(no_t{}, yes_t{}, (b & 1) ? (WNr >> 8) : WNr, END2, dummy);     /* the block's N bits in bits 0..7 of each half */
            }
            if ((m_enq >> b) & 1u) {
                const uint32_t d0 = pk_subs_vs(END2 + 0x00010001u, dup16(j0));
                const uint32_t ENDr = pk_subs_vs(END2, dup16(j0));
                const uint32_t WNr = wn(c);
                run8(yes_t{}, yes_t{}, (b & 1) ? (WNr >> 8) : WNr, ENDr, pk_nzmask(d0));
            }
            }
            if ((m_edge >> b) & 1u) {
                const uint32_t d0 = pk_subs_vs(END2 + 0x00010001u, dup16(j0));     /* mi of column j0 - 1 */
                const uint32_t ENDr = pk_subs_vs(END2, dup16(j0));      /* max(end - j0, 0): column constants stay block-relative */
                run8(yes_t{}, no_t{}, 0u, ENDr, pk_nzmask(d0));
            }
            /* fold the block: its row-max key into its 64-column group (column offsets only touch the low key bits), its
             * non-zero bits into the first / last trackers (one statement of seven packed ops, none reading its predecessor) */
            fold8<false>(mkg[g], Fnz, Lnz, mkb, nz8, (uint32_t)(j0 & 63) * 0x00010001u, (uint32_t)(j0 << 8) * 0x00010001u, k.ONE2);
        });

        L2_STAMP(3);
#if defined(BSW_L2_PRIO) && defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_s_setprio(BSW_L2_PRIO);
#endif
        /* row max over the groups, both seeds at once: the low byte of a group's key is the column inside the group
         * (< 64), so adding 64 g makes it absolute without touching the score byte */
        uint32_t mk2 = mkg[0];
        sfor<NG - 1>([&](auto gi) {
            constexpr int g = decltype(gi)::value + 1;
            mk2 = pk_max(mk2, mkg[g] + (uint32_t)(64 * g) * 0x00010001u);
        });
        /* ---- row tail (K7, K8), both seeds at once ---- */
        row_tail2<SYM>(p, k, i, r.ACT, h1, mk2, Fnz, Lnz);
    }
};


/* ---------------------------------------------------------------------------------------------------------------------
 * lane2l: the same two-seeds-per-lane arithmetic with the 8-column blocks walked by a RUN-TIME LOOP instead of unrolled
 * code.  The eh[] row does not live in a C++ array but behind a `ROW` accessor: on the GPU that is the accumulator
 * register file addressed through the VGPR index mode (s_set_gpr_idx_on: M0 offsets the AccVGPR number of
 * v_accvgpr_read/_write — bsw_lane2l_kernel.hip), on the CPU model a plain array.  One copy of each block body serves
 * every block, so the hot code is a few KB whatever the class width: the unrolled kernel's 232-column instantiation
 * spends half its time in instruction-cache misses at one wave per SIMD (profiles/r3/lane2_wide_icache.json).
 *   ROW::load8(b, T)            T[c] = eh[8b + c]
 *   ROW::store8(b, T)           eh[8b + c] = T[c]
 *   ROW::swap8w(b, T, Wc)       eh[8b + c] = T[c], then T[c] = eh[8b + 8 + c] and Wc = match bytes of block b + 1 (byte
 *                               (b+1) & 3 of the row's match words (b+1) >> 2, seed A's in [7:0], seed B's in [23:16]):
 *                               one statement, everything the next block needs under one index-mode window
 *   ROW::load8w(b, T, Wc)       T[c] = eh[8b + c] and block b's match bytes (the row's first block)
 *   ROW::put_rm(wd, a, b)       this row's match words (32 columns each) of seed A / B, word index static
 * Everything that was indexed by the block number at compile time is either block-relative already (key, bit and end
 * constants) or folded as it goes: the row max with the absolute column added per block, the first / last non-zero
 * column as a packed running minimum / maximum (K8). */
template <int QB, bool VM = false, bool SYM = true>
struct lane2l {
    using B = lane2<QB, VM, SYM>;
    static constexpr int QMAX = B::QMAX, NW = B::NW, NC = B::NC, KEEP_NONE = B::KEEP_NONE;

    struct state { pairv p; };

    /* K2 first row, closed form, block by block (sw_pe_array_sw_extend.v:1979,1957,1974) */
    template <class ROW>
    L2_MFN void init_row(const state &S, const consts &k, ROW &row)
    {
        const int h00 = half_of(S.p.H0, 0), h01 = half_of(S.p.H0, 1);
        for (int b = 0; b < QB; ++b) {
            uint32_t T[8];
            sfor<8>([&](auto ci) {
                constexpr int c = decltype(ci)::value;
                const int j = 8 * b + c;
                const int v0 = j == 0 ? h00 : imax(h00 - k.oe_ins - (j - 1) * k.e_ins, 0);
                const int v1 = j == 0 ? h01 : imax(h01 - k.oe_ins - (j - 1) * k.e_ins, 0);
                T[c] = pack2(v0, v1);
            });
            row.store8(b, T);
        }
    }

    /* the columns a band clamp dropped are zeroed (rare: only rows where the clamp moves some seed's beg) */
    template <class ROW>
    L2_MFN void zero_dropped(const rowp &r, const uni &u, ROW &row)
    {
        const bool bt0 = (r.BITE & 0xffffu) != 0, bt1 = (r.BITE >> 16) != 0;
        const int zlo0 = half_of(r.ZLO, 0), zlo1 = half_of(r.ZLO, 1), zhi0 = half_of(r.ZHI, 0), zhi1 = half_of(r.ZHI, 1);
        const int b0 = u.zl >> 3, b1 = (u.zh - 1) >> 3;
        for (int b = b0; b <= b1 && b < QB; ++b) {
            uint32_t T[8];
            row.load8(b, T);
            sfor<8>([&](auto ci) {
                constexpr int c = decltype(ci)::value;
                const int J = 8 * b + c;
                uint32_t keep = 0xffffffffu;
                if (bt0 && J >= zlo0 && J < zhi0) keep &= 0xffff0000u;
                if (bt1 && J >= zlo1 && J < zhi1) keep &= 0x0000ffffu;
                T[c] &= keep;
            });
            row.store8(b, T);
        }
    }

    /* One DP row for both seeds (see lane2::row_body for the arithmetic; the cell is the same function). */
    template <class QP, class KP, class WN, class ROW>
    L2_MFN void row_body(state &S, const consts &k, const int i, const rowp &r, const uni &u, const int (&tb)[2],
                         const QP &qp, const KP &kp, const WN &wn, ROW &row)
    {
        pairv &p = S.p;
        if (u.anybite) zero_dropped(r, u, row);
        {
            uint32_t rmA[NW], rmB[NW];
            B::match_words(qp, kp, 0, tb[0], (int)(p.BEG & 0xffffu), rmA);
            B::match_words(qp, kp, 1, tb[1], (int)(p.BEG >> 16), rmB);
            sfor<NW>([&](auto wi) { constexpr int wd = decltype(wi)::value; row.put_rm(wd, rmA[wd], rmB[wd]); });
        }
        const int pbA = tb[0] < 4 ? k.pb : k.pn, pbB = tb[1] < 4 ? k.pb : k.pn;
        const uint32_t Bv2 = pack2(pbA, pbB) << 8, D2 = pack2(pbA - k.pn, pbB - k.pn) << 8;
        const uint32_t END2 = p.END;
        const uint32_t h1c = pk_subs_vs(p.H0, dup16(imin(k.o_del + k.e_del * (i + 1), 0xffff)));     /* K4, both seeds at once */
        uint32_t h1 = pk_shl8(h1c & ~pk_nzmask(p.BEG)), f = 0;
        L2_STAMP(2);
        uint32_t mk2 = 0;                                   /* running row max: (m << 8) | absolute column, per half */
        uint32_t Fnz = 0xffffffffu, Lnz = 0;                /* K8: packed first / last non-zero column trackers */
        const int blo = u.jlo >> 3, bhi = imin(u.jhi >> 3, QB - 1), bem = u.jem >> 3;
#ifdef BSW_L2L_ALLNQ
        const uint32_t nblk = opaque_s(0xffffffffu);        /* (experiment: the query-N body in every block) */
#else
        const uint32_t nblk = opaque_s(u.nblk);
#endif
        uint32_t T[8], Wcur;
        uint32_t WNcur = nblk != 0 ? wn(blo >> 1) : 0u;     /* (waves with query Ns) the N planes of the chunk that holds the first block */
        row.load8w(blo, T, Wcur);
        /* one block: `edge` is a compile-time property of the loop it runs in (below), only the N test is per block */
        /* one block.  KIND 0: the plain body; 1: the ragged body (the row's first / last block, columns in order behind
         * scalar guards); 2: the query-N body (a wave that holds an N somewhere runs it in every block) */
        const auto step = [&](const int b, auto edge_c, auto kind_c) {
            constexpr bool EDGE = decltype(edge_c)::value;
            constexpr int KIND = decltype(kind_c)::value;
            const uint32_t j0 = 8u * (uint32_t)b;
            const uint32_t Wc = Wcur;
            uint32_t mkb = 0, nz8 = 0;
            const uint32_t J0d = j0 * 0x00010001u;
            uint32_t ENDx = END2, mi_in = 0;
            if (EDGE) {
                mi_in = pk_nzmask(pk_subs_vs(END2 + 0x00010001u, J0d));     /* mi of column j0 - 1 */
                ENDx = pk_subs_vs(END2, J0d);
            }
            if constexpr (KIND == 0) B::template block8<EDGE, false>(T, Wc, 0u, Bv2, D2, k, ENDx, mi_in, h1, f, mkb, nz8);
            if constexpr (KIND == 1) B::template block8_seq<EDGE>(T, Wc, Bv2, k, ENDx, mi_in, EDGE ? (u.jhi & 7) : (u.jlo & 7), h1, f, mkb, nz8);
            if constexpr (KIND == 2) {
                /* the query-N body in EVERY block of a wave that holds an N anywhere, no test per block: with zero N bits it
                 * computes what the plain body does.  (Round 3 tested per block and ran the plain body where it could: a wave
                 * of 128 N-holding queries has an N in 23 of its 29 blocks, and the tested loops cost every block 16 % —
                 * such waves ran 1.31x as long as plain ones; they start first, and at one wave per SIMD and three rounds per
                 * launch the longest waves set the launch's length: 250 bp lost 8.6 % to 8.6 % of its waves,
                 * profiles/r4/lane2l_n_waves.json.)  The N bits of the NEXT block are fetched before this block's body. */
                const uint32_t WNc = (b & 1) ? (WNcur >> 8) : WNcur;        /* the block's N bits in bits 0..7 of each half */
                WNcur = wn(imin(b + 1, QB - 1) >> 1);
                B::template block8<EDGE, true>(T, Wc, WNc, Bv2, D2, k, ENDx, mi_in, h1, f, mkb, nz8);
            }
            /* row max: the block's key carries the column inside the block; + j0 makes it absolute (< 256: low byte).
             * K8: nz8 = non-zero bits of the block's stored eh entries, seed A in [7:0], seed B in [23:16].
             * last:  key = (j0 << 8) + bits for a non-empty block, 0 for an empty one: the maximum keeps the highest block;
             * first: key - 1 (no borrow out of the low byte for a non-empty block; 0xffff for an empty one, which never
             *        wins): the minimum keeps the lowest block, and the tail reads its bits back as (low byte) + 1.
             * (Folding the bits per 32-column word as the unrolled kernel does, with the match words shifted along, was
             * slower here: 1508 vs 1579 GCUPS on the 250 bp workload.) */
            fold8(mk2, Fnz, Lnz, mkb, nz8, J0d, J0d << 8, k.ONE2);
            row.swap8w(b, T, Wcur);                          /* eh[8b ..] <- T, T <- eh[8b + 8 ..], match bytes of b + 1 */
        };
        using dense_t = std::integral_constant<bool, false>;
        using edge_t = std::integral_constant<bool, true>;
        using plain_t = std::integral_constant<int, 0>;
        using ragged_t = std::integral_constant<int, 1>;
        using allnq_t = std::integral_constant<int, 2>;
        /* the blocks below every active seed's `end` first (mask-free bodies), then the ones that hold some seed's `end`:
         * two loops, so that the dense / edge decision costs no scalar instructions per block (at one wave per SIMD the
         * scalar instructions of the block loop take issue slots like everything else).  A wave whose queries hold no N
         * (the usual case) runs loops without any test per block, the ragged first / last block peeled off them. */
        int b = blo;
        const int bd = imin(bem - 1, bhi);
#ifdef BSW_L2L_NOFAST
        if (false) {
#else
        if (nblk == 0) {
#endif
            if ((u.jlo & 7) != 0 && b <= bd) { step(b, dense_t{}, ragged_t{}); ++b; }
            for (; b <= bd; ++b) step(b, dense_t{}, plain_t{});
            const bool rl = (u.jhi & 7) != 7;
            const int be = rl ? bhi - 1 : bhi;
            for (; b <= be; ++b) step(b, edge_t{}, plain_t{});
            if (rl && b <= bhi) step(b, edge_t{}, ragged_t{});
        } else {
            for (; b <= bd; ++b) step(b, dense_t{}, allnq_t{});
            for (; b <= bhi; ++b) step(b, edge_t{}, allnq_t{});
        }
        L2_STAMP(3);
        row_tail2<SYM>(p, k, i, r.ACT, h1, mk2, Fnz, Lnz);        /* K7, K8 for both seeds at once */
    }
};


/* ---------------------------------------------------------------------------------------------------------------------
 * lane2g: the same packed arithmetic with a seed PAIR spread over a GROUP of eight lanes — the kernel between the general
 * ones (one / four seeds per wavefront, int32) and the lane kernels (128 seeds per wavefront, one lane walks a whole row:
 * a launch lasts one wave's lifetime, ~1 ms per 131-column side, however few seeds it holds).  Here a wavefront holds 16
 * seeds and lives a tenth as long (bsw_lane2g_kernel.hip: batches of a few thousand to ~100 k seeds, wire-format groups).
 *   Columns are STRIPED: lane g of the group owns the 8-column block j0 = 64 s + 8 g of stripe s = 0 .. NS-1, in registers
 *   for the whole extension (T[s][c] = eh[j0 + c], both seeds packed as in lane2).  A row costs the stripes between the
 *   smallest beg and the largest end of the wavefront's seeds.
 *   The columns of a row depend on each other through f (the horizontal gap) and through H(i, j-1) in the stored pair:
 *   - f: a lane first works out what its block OFFERS the columns right of it with nothing coming in (phase_a: the f
 *     recurrence over max(M, e) — what f itself adds to h never matters to the next f, because o_ins >= 0), the offers
 *     are combined across the group by a max-plus prefix scan with a decay of 8 e_ins per lane (three DPP steps), and
 *     the block body then runs ONCE with the true f entering it (phase_b = lane2::block8, the very cell of the lane kernels);
 *   - H(i, j0 - 1), needed only for the stored eh[j0] of the NEXT row, is the left neighbour's last h of THIS row: the body
 *     runs with 0 in its place and phase_c puts the true byte (and the column's non-zero bit) in afterwards.
 *   Everything per seed (pairv, K3 / K7 / K8) is computed redundantly by the eight lanes of the group; the row maximum, the
 *   first / last non-zero trackers and eh[end].h are reduced over the group by three-step DPP butterflies.
 * The phases are per-lane functions so that the CPU model of the tests (tests/lane2_model.cpp) runs THIS code with the
 * exchanges between them restated as array shifts. */
template <int NS, bool VM = false, bool SYM = true>
struct lane2g {
    using B = lane2<8 * NS, VM, SYM>;
    static constexpr int G = 8, COLS = 64 * NS;

    struct state {
        uint32_t T[NS][8];              /* T[s][c] = eh[64 s + 8 g + c], half = {e:8 | h:8} */
        pairv p;
    };
    struct rowk {                       /* per-lane values of the current row (equal in the lanes of a group) */
        uint32_t Bv2, D2;               /* mismatch / N penalties of this row's target bases, scaled (lane2::row_body) */
        uint32_t tnm8;                  /* 0x00ff in a half whose target base is not an N (an N row has no match bits) */
        uint32_t sel;                   /* byte selector of the two seeds' match bytes for this row's target bases */
        uint32_t h1init;                /* K4: H(i, -1) scaled, 0 unless beg == 0 */
        uint32_t BEG2, END2;
    };
    struct stripe_in { uint32_t Wc, WN, ENDr, mi_in, J0d; };

    L2_MFN uint32_t j0_of(int s, int g) { return (uint32_t)(64 * s + 8 * g); }

    /* K2 first row, closed form (sw_pe_array_sw_extend.v:1979,1957,1974) */
    L2_MFN void init_row(state &S, const consts &k, const int g)
    {
        const int h00 = half_of(S.p.H0, 0), h01 = half_of(S.p.H0, 1);
        sfor<NS>([&](auto si) {
            constexpr int s = decltype(si)::value;
            sfor<8>([&](auto ci) {
                constexpr int c = decltype(ci)::value;
                const int j = 64 * s + 8 * g + c;
                const int v0 = j == 0 ? h00 : imax(h00 - k.oe_ins - (j - 1) * k.e_ins, 0);
                const int v1 = j == 0 ? h01 : imax(h01 - k.oe_ins - (j - 1) * k.e_ins, 0);
                S.T[s][c] = pack2(v0, v1);
            });
        });
    }

    /* the columns [zlo, zhi) a band clamp dropped are zeroed (rare), keeping "eh[j] == 0 below beg" */
    L2_MFN void zero_dropped(state &S, const rowp &r, const int g)
    {
        sfor<NS>([&](auto si) {
            constexpr int s = decltype(si)::value;
            sfor<8>([&](auto ci) {
                constexpr int c = decltype(ci)::value;
                const uint32_t Jd = dup16(64 * s + 8 * g + c);
                const uint32_t below_hi = pk_nzmask(pk_subs(r.ZHI, Jd));          /* J < zhi */
                const uint32_t below_lo = pk_nzmask(pk_subs(r.ZLO, Jd));          /* J < zlo */
                S.T[s][c] &= ~(r.BITE & below_hi & ~below_lo);
            });
        });
    }

    L2_MFN rowk row_consts(const pairv &p, const consts &k, const int i, const int (&tb)[2])
    {
        rowk rk;
        const int pbA = tb[0] < 4 ? k.pb : k.pn, pbB = tb[1] < 4 ? k.pb : k.pn;
        rk.Bv2 = pack2(pbA, pbB) << 8;
        rk.D2 = pack2(pbA - k.pn, pbB - k.pn) << 8;
        rk.tnm8 = (tb[0] < 4 ? 0x000000ffu : 0u) | (tb[1] < 4 ? 0x00ff0000u : 0u);
        rk.sel = 0x0c000c00u | (uint32_t)(tb[0] & 3) | ((uint32_t)(4 + (tb[1] & 3)) << 16);
        const uint32_t h1c = pk_subs_vs(p.H0, dup16(imin(k.o_del + k.e_del * (i + 1), 0xffff)));     /* K4, both seeds at once */
        rk.h1init = pk_shl8(h1c & ~pk_nzmask(p.BEG));
        rk.BEG2 = p.BEG; rk.END2 = p.END;
        return rk;
    }

    /* does some seed of this lane need stripe s this row?  (columns [beg, end] — the column `end` takes a store) */
    L2_MFN uint32_t needs_stripe(const rowk &rk, const uint32_t act, const int s)
    {
        const uint32_t left = pk_nzmask(pk_subs_sv(dup16(64 * s + 64), rk.BEG2));             /* beg < 64 (s + 1) */
        const uint32_t right = pk_nzmask(pk_subs_vs(rk.END2 + 0x00010001u, dup16(64 * s)));   /* end + 1 > 64 s */
        return act & left & right;
    }

    /* Phase A: the block's operands and its OFFER — f at the lane's exit when nothing enters it.  mA / mB: the four per-base
     * match bytes of the block's columns (seed A / B), WNs: their query-N bits (A in [7:0], B in [23:16]). */
    template <bool NQ>
    L2_MFN uint32_t phase_a(const uint32_t (&T)[8], const rowk &rk, const consts &k, const uint32_t mA, const uint32_t mB,
                            const uint32_t WNs, const uint32_t J0d, stripe_in &si)
    {
        /* the columns below beg lose their match bits: keep = bits [clamp(beg - j0, 0, 8), 8) */
        const uint32_t sh = pk_min_vs(pk_subs(rk.BEG2, J0d), 0x00080008u);
        const uint32_t keep = pk_shlv(sh, 0x00ff00ffu) & 0x00ff00ffu;
        si.Wc = byte_pair_sel(mA, mB, rk.sel) & keep & rk.tnm8;
        si.WN = WNs;
        si.J0d = J0d;
        si.ENDr = pk_subs(rk.END2, J0d);                                   /* max(end - j0, 0): the bodies' masks are block-relative */
        si.mi_in = pk_nzmask(pk_subs(rk.END2 + 0x00010001u, J0d));         /* 0xffff where j0 <= end: the mask of column j0 - 1 */
        uint32_t f = 0;
        sfor<8>([&](auto ci) {
            constexpr int c = decltype(ci)::value;
            constexpr uint32_t BIT = (uint32_t)(1u << c) * 0x00010001u;
            const uint32_t hd = pk_shl8(T[c]);
            uint32_t X = pk_mad_vsv(si.Wc & BIT, k.MC[c], hd);
            if (NQ) X = pk_mad((c ? (WNs >> c) : WNs) & 0x00010001u, rk.D2, X);
            uint32_t M = pk_subs(X, rk.Bv2);
            if (VM) M &= pk_nzmask(hd);
            const uint32_t gsrc = VM ? M : pk_max(M, T[c]);                 /* what a gap opens from, without f (see above) */
            const uint32_t tI = pk_subs_vs(gsrc, SYM ? k.OED2s : k.OEI2s);
            f = pk_max(pk_subs_vs(f, SYM ? k.ED2s : k.EI2s), tI);
        });
        return f;
    }

    /* Phase B: the block with the true f entering it (lane2's body, masks relative to `end`); hl = H of the lane's last
     * column below `end` (0: it has none), f = what leaves the lane */
    template <bool NQ>
    L2_MFN void phase_b(uint32_t (&T)[8], const stripe_in &si, const rowk &rk, const consts &k, uint32_t &hl, uint32_t &f, uint32_t &mkb, uint32_t &nz8)
    {
        hl = 0;
        B::template block8<true, NQ>(T, si.Wc, si.WN, rk.Bv2, rk.D2, k, si.ENDr, si.mi_in, hl, f, mkb, nz8);
    }

    /* Phase C: eh[j0].h = H(i, j0 - 1) from the left neighbour (column j0 takes a store iff j0 <= end), and the column's
     * non-zero bit with it */
    L2_MFN void phase_c(uint32_t &T0, uint32_t &nz8, const stripe_in &si, const consts &k, const uint32_t Hin)
    {
        T0 = bfi(si.mi_in & 0x00ff00ffu, pk_shr8(Hin), T0);
        nz8 = (nz8 & 0xfffefffeu) | pk_min_vs(T0 & si.mi_in, k.ONE2);
    }

    /* the block's row-max key and non-zero bits into the lane's running values (lane2l's fold8 with the column in a VGPR) */
    L2_MFN void foldv(uint32_t &mk2, uint32_t &Fnz, uint32_t &Lnz, const uint32_t mkb, const uint32_t nz8, const uint32_t J0d, const consts &k)
    {
        const uint32_t key = pk_mad(pk_min_vs(nz8, k.ONE2), pk_shl8(J0d), nz8);
        mk2 = pk_max(mk2, mkb + J0d);
        Lnz = pk_max(Lnz, key);
        Fnz = pk_min(Fnz, pk_sub_vs(key, k.ONE2));
    }

    /* eh[end].h comes from the lane that holds column end - 1: 1 <= end - j0 <= 8 */
    L2_MFN uint32_t hfin_cand(const stripe_in &si, const uint32_t hl)
    {
        return hl & pk_nzmask(si.ENDr) & ~pk_nzmask(pk_subs_vs(si.ENDr, 0x00080008u));
    }
    /* ... and with an empty range at column 0 it is K4's value (lane2: h1 keeps its initial value) */
    L2_MFN uint32_t hfin_of(const rowk &rk, const uint32_t reduced)
    {
        return bfi(pk_nzmask(rk.END2), reduced, rk.h1init);
    }
};

}  // namespace l2
}  // namespace bsw
#endif
