/*
 * bsw_align_kernel.hip — bwa's striped local alignment ksw_align2 (ksw_u8 / ksw_i16; SURVEY.md §8f row F4, second
 * half: the Smith-Waterman of mate rescue, mem_matesw) on gfx950.
 *
 * The outputs of the SSE2 code depend on its striping (which query position shares a vector with which, where the
 * lazy-F loop stops, that E is taken from H before the lazy-F correction, the memory order of the qe search — see
 * oracle/ksw_align_ref.c), so the kernel keeps the striping: ONE 16-LANE DPP ROW PLAYS ONE __m128i.  Lane l of the
 * row is byte lane l (8-bit mode: one alignment per row), or the row holds two alignments of 8 word lanes each
 * (16-bit mode); the slen vectors of H, E and Hmax are slen registers per lane, statically addressed (H is updated in
 * place: the old value is the next column's diagonal); _mm_slli_si128 is one `row_shr:1` with the group's lane 0
 * forced to zero; the horizontal max and the "all lanes done" test of the lazy-F loop are DPP butterflies / a ballot.
 * Four (eight) alignments per wavefront; the per-lane score profile lives in LDS, one column per thread.  The sub-optimal list b[] goes to HBM (8 bytes per
 * qualifying row) and is scanned by the row's 16 lanes afterwards.  The start-point pass (KSW_XSTART) runs in the same
 * kernel on the mirrored prefixes by index arithmetic, no sequence is reversed in memory.
 * Integer max/add work, ~14 VALU + 1 LDS read per cell and lane: VALU-issue-bound like the extension kernels, but at
 * one cell per lane-op (no two-per-register packing: the striping decides what shares a register).
 */
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>

#include "bsw_device.h"
#include "bsw_stage.h"

namespace bsw {

namespace {

#define A_XBYTE  0x10000
#define A_XSTOP  0x20000
#define A_XSUBO  0x40000
#define A_XSTART 0x80000

template <int CTRL>
__device__ __forceinline__ int adpp(int old, int src)
{
    return __builtin_amdgcn_update_dpp(old, src, CTRL, 0xf, 0xf, false);
}
/* max over the GW (16 or 8) lanes of a group, in every lane: (mirror,) half mirror, two quad permutes */
template <int GW>
__device__ __forceinline__ int row_max16(int v)
{
    if (GW == 16) v = max(v, adpp<0x140>(INT_MIN, v));   /* row_mirror      */
    v = max(v, adpp<0x141>(INT_MIN, v));            /* row_half_mirror */
    v = max(v, adpp<0x4e>(INT_MIN, v));             /* quad_perm [2,3,0,1] */
    v = max(v, adpp<0xb1>(INT_MIN, v));             /* quad_perm [1,0,3,2] */
    return v;
}
template <int GW>
__device__ __forceinline__ unsigned long long row_max16_u64(unsigned long long v)
{
    /* two 32-bit butterflies would not keep (hi, lo) together: compare as 64 bits after moving both halves */
    for (int s = GW == 16 ? 0 : 1; s < 4; ++s) {
        int lo = (int)(uint32_t)v, hi = (int)(uint32_t)(v >> 32), lo2, hi2;
        switch (s) {
        case 0: lo2 = adpp<0x140>(0, lo); hi2 = adpp<0x140>(0, hi); break;
        case 1: lo2 = adpp<0x141>(0, lo); hi2 = adpp<0x141>(0, hi); break;
        case 2: lo2 = adpp<0x4e>(0, lo); hi2 = adpp<0x4e>(0, hi); break;
        default: lo2 = adpp<0xb1>(0, lo); hi2 = adpp<0xb1>(0, hi); break;
        }
        const unsigned long long o = ((unsigned long long)(uint32_t)hi2 << 32) | (uint32_t)lo2;
        v = o > v ? o : v;
    }
    return v;
}
__device__ __forceinline__ int base_at(const uint64_t *__restrict__ seq, uint32_t off, int idx)
{
    const int c = (int)((seq[off + (uint32_t)(idx >> 4)] >> ((idx & 15) * 4)) & 7ull);
    return c > 4 ? 4 : c;
}

struct akswr { int score, te, qe, score2, te2, tb, qb; };

/* One run of ksw_u8 (BYTE) / ksw_i16 for the alignment of this 16-lane row.  `on`: the row takes part (row-uniform).
 * Query position k is q[qrev ? qlast - k : k]; target row i is t[i <= trev ? trev - i : i] (trev = -1: forwards). */
template <int SLEN, bool BYTE>
__device__ __forceinline__ akswr align_pass(const bool on, const int qlen, const int tlen, const uint64_t *__restrict__ seq,
                                            const uint32_t q_off, const int qlast, const bool qrev, const uint32_t t_off, const int trev,
                                            const int xtra, const bsw_dparams &P, const int shift, const int mx,
                                            int8_t (*prof)[SLEN][256], unsigned long long *__restrict__ bl)
{
    constexpr int NP = BYTE ? 16 : 8;                /* lanes of the SSE vector = lanes of the group: a 16-lane DPP row holds one
                                                         8-bit alignment or two 16-bit ones */
    const int tid = threadIdx.x, l = tid & (NP - 1);
    constexpr bool lane_on = true;
    const int slen = (qlen + NP - 1) / NP;
    const int oe_del = P.o_del + P.e_del, oe_ins = P.o_ins + P.e_ins, e_del = P.e_del, e_ins = P.e_ins;
    const int hcap = BYTE ? 255 - shift : 32767;
    const int minsc = (xtra & A_XSUBO) ? (xtra & 0xffff) : 0x10000;
    const int endsc = (xtra & A_XSTOP) ? (xtra & 0xffff) : 0x10000;
    /* ksw_qinit: the score of (target base a, query position j + l * slen), 0 past the end of the query */
#pragma unroll
    for (int j = 0; j < SLEN; ++j) {
        const int k = j + l * slen;
        const bool real = on && lane_on && j < slen && k < qlen;
        const int qc = real ? base_at(seq, q_off, qrev ? qlast - k : k) : 0;
#pragma unroll
        for (int a = 0; a < 5; ++a) prof[a][j][tid] = real ? P.mat[a * 5 + qc] : (int8_t)0;
    }
    int H[SLEN], E[SLEN], Hmax[SLEN];
#pragma unroll
    for (int j = 0; j < SLEN; ++j) H[j] = E[j] = Hmax[j] = 0;
    int gmax = 0, te = -1, n_b = 0, last_i = -2, last_s = 0, hl = 0;       /* hl = H[slen - 1] of the previous row */
    bool live = on && slen > 0;
    for (int i = 0;; ++i) {
        const bool ra = live && i < tlen;
        if (__builtin_amdgcn_ballot_w64(ra) == 0) break;
        const int ti = ra ? (i <= trev ? trev - i : i) : 0;
        const int tb = ra ? base_at(seq, t_off, ti) : 0;
        int f = 0, mxv = 0;
        int h = adpp<0x111>(0, hl);                                        /* H(i-1,-1): the last vector, one lane up */
        h = l == 0 ? 0 : h;
#pragma unroll
        for (int j = 0; j < SLEN; ++j) {
            if (ra && j < slen) {
                const int s = prof[tb][j][tid];
                int hh = min(h + s, hcap);
                if (BYTE) hh = max(hh, 0);
                const int e = E[j];
                hh = max(max(hh, e), f);
                mxv = max(mxv, hh);
                h = H[j];
                H[j] = hh;
                E[j] = max(max(e - e_del, hh - oe_del), 0);
                f = max(max(f - e_ins, hh - oe_ins), 0);
            }
        }
        /* lazy F: at most 16 rounds, out as soon as no lane's F exceeds H - oe_ins */
        bool lz = ra;
        for (int k = 0; k < 16; ++k) {
            if (__builtin_amdgcn_ballot_w64(lz) == 0) break;
            const int fs = adpp<0x111>(0, f);
            f = lz ? (l == 0 ? 0 : fs) : f;
#pragma unroll
            for (int j = 0; j < SLEN; ++j) {
                if (__builtin_amdgcn_ballot_w64(lz) == 0) break;           /* usually after the first vector or two */
                const bool go = lz && j < slen;
                int hh = max(H[j], f);
                H[j] = go ? hh : H[j];
                hh = max(hh - oe_ins, 0);
                const int fn = max(f - e_ins, 0);
                f = go ? fn : f;
                const unsigned long long m = __builtin_amdgcn_ballot_w64(go && lane_on && fn > hh);
                const bool any = ((m >> (tid & (64 - NP))) & (NP == 16 ? 0xffffull : 0xffull)) != 0;
                lz = go ? any : lz;                                        /* (rows past their slen keep lz and skip on their own) */
            }
        }
#pragma unroll
        for (int j = 0; j < SLEN; ++j) hl = (ra && j == slen - 1) ? H[j] : hl;
        const int imax = row_max16<NP>(mxv);
        if (ra && imax >= minsc) {                                         /* the b array of sub-optimal ends */
            if (n_b == 0 || last_i + 1 != i) {
                if (l == 0) bl[n_b] = ((unsigned long long)(uint32_t)imax << 32) | (uint32_t)i;
                ++n_b; last_i = i; last_s = imax;
            } else if (last_s < imax) {
                if (l == 0) bl[n_b - 1] = ((unsigned long long)(uint32_t)imax << 32) | (uint32_t)i;
                last_i = i; last_s = imax;
            }
        }
        if (ra && imax > gmax) {
            gmax = imax; te = i;
#pragma unroll
            for (int j = 0; j < SLEN; ++j) Hmax[j] = H[j];
            if (BYTE ? (gmax + shift >= 255 || gmax >= endsc) : (gmax >= endsc)) live = false;
        }
    }
    akswr r;
    r.score = BYTE ? (gmax + shift < 255 ? gmax : 255) : gmax;
    r.te = te; r.qe = -1; r.score2 = -1; r.te2 = -1; r.tb = -1; r.qb = -1;
    if (!BYTE || r.score != 255) {
        /* qe: the first maximum of the kept column in memory order i = j * NP + lane -> position j + lane * slen */
        int key = -1;
#pragma unroll
        for (int j = 0; j < SLEN; ++j)
            if (lane_on && j < slen) key = max(key, (Hmax[j] << 16) | (0xffff - (j * NP + l)));
        key = row_max16<NP>(key);
        if (slen > 0) {
            const int mi = 0xffff - (key & 0xffff);
            r.qe = mi / NP + (mi % NP) * slen;
        }
        if (n_b > 0) {                                                     /* second best: the first strictly larger entry outside [low, high] wins */
            __threadfence_block();
            const int d = (r.score + mx - 1) / mx, low = te - d, high = te + d;
            unsigned long long best = 0;                                   /* (score + 1) << 32 | ~index : 0 = none */
            for (int x = l; x < n_b; x += NP) {
                const unsigned long long ent = bl[x];
                const int e = (int)(uint32_t)ent, sc = (int)(ent >> 32);
                if (e < low || e > high) {
                    const unsigned long long kk = ((unsigned long long)(uint32_t)(sc + 1) << 32) | (uint32_t)(0x7fffffff - x);
                    best = kk > best ? kk : best;
                }
            }
            best = row_max16_u64<NP>(best);
            if (best != 0) {
                const int x = 0x7fffffff - (int)(uint32_t)best;
                const int sc = (int)(best >> 32) - 1;
                if (sc > r.score2) { r.score2 = sc; r.te2 = (int)(uint32_t)bl[x]; }
            }
        }
    }
    return r;
}

}  // namespace

/* one local alignment per 16-lane row; tasks[] says where its sequences sit in seq, its flags, its slice of the b[] scratch */
template <int SLEN, bool BYTE>
__global__ __launch_bounds__(256) void bsw_align_kernel(const bsw_dparams P, const uint64_t *__restrict__ seq,
                                                        const bsw_adtask *__restrict__ tasks, const uint32_t *__restrict__ order,
                                                        const uint32_t n, unsigned long long *__restrict__ blist, bsw_kswr *__restrict__ out)
{
    __shared__ int8_t prof[5][SLEN][256];
    constexpr int GW = BYTE ? 16 : 8, APB = 256 / GW;               /* lanes per alignment, alignments per workgroup */
    const int tid = threadIdx.x, l = tid & (GW - 1);
    const uint32_t slot = blockIdx.x * (uint32_t)APB + (uint32_t)(tid / GW);
    const bool valid = slot < n;
    const uint32_t ai = order[valid ? slot : 0];
    const bsw_adtask T = tasks[ai];
    int smin = 127, smax = 0;
#pragma unroll
    for (int a = 0; a < 25; ++a) { smin = min(smin, (int)P.mat[a]); smax = max(smax, (int)P.mat[a]); }
    const int shift = (256 - (smin & 0xff)) & 0xff, mx = smax;
    unsigned long long *bl = blist + T.b_off;
    akswr r = align_pass<SLEN, BYTE>(valid, T.qlen, T.tlen, seq, T.q_off, 0, false, T.t_off, -1, T.xtra, P, shift, mx, prof, bl);
    const bool second = valid && !((T.xtra & A_XSTART) == 0 || ((T.xtra & A_XSUBO) && r.score < (T.xtra & 0xffff)));
    if (__builtin_amdgcn_ballot_w64(second) != 0) {
        /* (the profile is rebuilt for the mirrored prefix; every thread owns its column of it: no barrier) */
        const akswr rr = align_pass<SLEN, BYTE>(second, r.qe + 1, T.tlen, seq, T.q_off, r.qe, true, T.t_off, r.te, A_XSTOP | r.score,
                                                P, shift, mx, prof, bl);
        if (second && r.score == rr.score) { r.tb = r.te - rr.te; r.qb = r.qe - rr.qe; }
    }
    if (valid && l == 0) {
        bsw_kswr o;
        o.score = r.score; o.te = r.te; o.qe = r.qe; o.score2 = r.score2; o.te2 = r.te2; o.tb = r.tb; o.qb = r.qb;
        out[ai] = o;
    }
}

/* classes: (mode, vectors per lane) -> query length up to lanes * vectors */
/* (10 / 20 vectors: 150 bp reads, the common case, without the predicated tail of the 16 / 32 vector kernels) */
/* (32 / 64 byte vectors and 64 / 128 word vectors: queries up to 1024 bases — bwa's ksw_align2 has no length limit and
 * mate rescue of 2x300 reads must not lose its alignment; the long classes run at one wave per SIMD, rare by design) */
static const struct { int byte, slen; } kAlignClasses[] = {{1, 8}, {1, 10}, {1, 16}, {1, 32}, {1, 64}, {0, 16}, {0, 20}, {0, 32}, {0, 64}, {0, 128}};
int align_class_count() { return (int)(sizeof(kAlignClasses) / sizeof(kAlignClasses[0])); }
int align_class_of(int qlen, int byte_mode)
{
    for (int c = 0; c < align_class_count(); ++c)
        if (kAlignClasses[c].byte == (byte_mode ? 1 : 0) && qlen <= kAlignClasses[c].slen * (byte_mode ? 16 : 8)) return c;
    return -1;
}

hipError_t launch_align(int cls, const bsw_dparams &P, const uint64_t *seq, const bsw_adtask *tasks, const uint32_t *order, uint32_t n,
                        unsigned long long *blist, bsw_kswr *out, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    const bool byte = kAlignClasses[cls].byte != 0;
    const uint32_t apb = byte ? 16u : 32u;
    const dim3 grid((n + apb - 1u) / apb), block(256);
    switch (cls) {
    case 0: hipLaunchKernelGGL((bsw_align_kernel<8, true>), grid, block, 0, s, P, seq, tasks, order, n, blist, out); break;
    case 1: hipLaunchKernelGGL((bsw_align_kernel<10, true>), grid, block, 0, s, P, seq, tasks, order, n, blist, out); break;
    case 2: hipLaunchKernelGGL((bsw_align_kernel<16, true>), grid, block, 0, s, P, seq, tasks, order, n, blist, out); break;
    case 3: hipLaunchKernelGGL((bsw_align_kernel<32, true>), grid, block, 0, s, P, seq, tasks, order, n, blist, out); break;
    case 4: hipLaunchKernelGGL((bsw_align_kernel<64, true>), grid, block, 0, s, P, seq, tasks, order, n, blist, out); break;
    case 5: hipLaunchKernelGGL((bsw_align_kernel<16, false>), grid, block, 0, s, P, seq, tasks, order, n, blist, out); break;
    case 6: hipLaunchKernelGGL((bsw_align_kernel<20, false>), grid, block, 0, s, P, seq, tasks, order, n, blist, out); break;
    case 7: hipLaunchKernelGGL((bsw_align_kernel<32, false>), grid, block, 0, s, P, seq, tasks, order, n, blist, out); break;
    case 8: hipLaunchKernelGGL((bsw_align_kernel<64, false>), grid, block, 0, s, P, seq, tasks, order, n, blist, out); break;
    case 9: hipLaunchKernelGGL((bsw_align_kernel<128, false>), grid, block, 0, s, P, seq, tasks, order, n, blist, out); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace bsw
