/*
 * bsw_stage_kernel.hip — device side of the batch manager: everything between "the caller's bytes
 * have landed in HBM" and "the DP kernels can start".
 *
 * The reference's batch manager moves 256 KiB task batches into the PE arrays without a CPU in the
 * loop (batch_manager.v:358-739, tbb.v:12-212) and its task parser hands tasks to PEs as they come
 * (sw_pe_array_task_parse.v:1600-1648).  Here the host only validates lengths and DMAs the raw
 * bytes; these kernels do the rest:
 *   bsw_pack_kernel      byte-per-base -> 16 bases per uint64 (codes > 4 become 4 = N)
 *   bsw_bin_count/scan/scatter   counting sort of the seeds into kernel bins: wave-per-task classes
 *                        by eh[] columns per lane, lane bins per side by (class, query with /
 *                        without an N, query length descending) — BASELINE.json's "(qlen, tlen,
 *                        band-width) bins"; the N key keeps the wavefronts of the two-seeds-per-
 *                        lane kernels free of the query-N block bodies (a wave runs them for a
 *                        block as soon as ONE of its 128 queries has an N there)
 *   bsw_wire_pack_kernel the reference's 256 KiB wire format (8 bases per 32-bit word, first base in
 *                        bits [31:28], one nibble stream per task: proc_element.v:1638,1677) -> seq
 * All three are HBM-bound byte/index work (no MFMA, nothing to tile): coalesced dword loads,
 * v_alignbyte for the unaligned starts, LDS-privatised histograms.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bsw_device.h"
#include "bsw_stage.h"

namespace bsw {

/* 8 base bytes (codes 0..4) -> 8 nibbles in the low 32 bits */
__device__ __forceinline__ uint64_t squeeze8(uint64_t x)
{
    x = (x | (x >> 4)) & 0x00FF00FF00FF00FFull;
    x = (x | (x >> 8)) & 0x0000FFFF0000FFFFull;
    x = (x | (x >> 16)) & 0x00000000FFFFFFFFull;
    return x;
}

/* codes 5..7 -> 4; bytes >= 8 are clamped bytewise (never produced by bwa) */
__device__ __forceinline__ uint64_t clamp_codes(uint64_t x)
{
    if (x & 0xF8F8F8F8F8F8F8F8ull) {
        uint64_t v = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint64_t b = (x >> (8 * k)) & 0xff;
            v |= (b > 4 ? 4ull : b) << (8 * k);
        }
        return v;
    }
    const uint64_t n = x & 0x0404040404040404ull;
    return x & ~((n >> 1) | (n >> 2));
}

/* ---- targets out of the device-resident 2-bit reference (replaces the host's bns_get_seq +
 * reversal so that only the read crosses PCIe).  Coordinates are bwa's: x in [0, l_pac) is the forward strand,
 * x in [l_pac, 2*l_pac) the reverse complement: base(x) = 3 - pac[2*l_pac - 1 - x].  pac packs 4 bases per byte,
 * first base in the top two bits. ---- */

/* the 16 bases at pac positions pos .. pos+15 as 32 big-endian bits (base pos in bits [31:30]).  pos may run off
 * either end of the pac by up to 15 bases: the byte index is clamped, the caller masks those bases away */
__device__ __forceinline__ uint32_t pac16(const uint8_t *__restrict__ pac, const int64_t last_byte, const int64_t pos)
{
    const int64_t b = pos >> 2;
    uint64_t v = 0;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        int64_t i = b + k;
        i = i < 0 ? 0 : (i > last_byte ? last_byte : i);
        v = (v << 8) | pac[i];
    }
    return (uint32_t)(v >> (8 - 2 * (int)(pos & 3)));
}

/* 16 two-bit bases (base k in bits [2k, 2k+1]) -> 16 nibbles */
__device__ __forceinline__ uint64_t spread16(uint32_t u)
{
    uint64_t x = u;
    x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
    x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
    x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x << 2)) & 0x3333333333333333ull;
    return x;
}

/* word w of a target of tlen bases whose base i sits at coordinate x0 + dir*i (one strand: the host rejects windows
 * that bridge l_pac) */
__device__ __forceinline__ uint64_t fetch_word(const uint8_t *__restrict__ pac, const int64_t l_pac, const int64_t x0, const int dir,
                                               const int tlen, const int w)
{
    const bool fwd = x0 < l_pac;
    const int64_t p0 = fwd ? x0 : (l_pac << 1) - 1 - x0;      /* pac position of base 0 */
    const bool up = fwd == (dir > 0);                         /* pac positions ascend with i */
    const int64_t last_byte = ((l_pac + 3) >> 2) - 1;
    uint32_t u;
    if (up) {
        u = __builtin_bitreverse32(pac16(pac, last_byte, p0 + 16 * (int64_t)w));
        u = ((u >> 1) & 0x55555555u) | ((u & 0x55555555u) << 1);
    } else
        u = pac16(pac, last_byte, p0 - 16 * (int64_t)w - 15); /* base 0 of the word is the LAST of the 16: already little-endian */
    uint64_t v = spread16(u);
    if (!fwd) v ^= 0x3333333333333333ull;
    const int valid = tlen - 16 * w;
    if (valid < 16) v &= (1ull << (4 * valid)) - 1ull;
    return v;
}

/* 16 lanes per sequence, 4 sequences per seed (leftQ, leftT, rightQ, rightT); lane k packs words k, k+16, ...
 * rev_left: the left query is read BACKWARDS from its offset (base k = raw[off - k]): a read DMA'd as it is holds
 * query[0..qbeg) forwards, mem_chain2aln extends it reversed (the host's reversal loop, done here for free).
 * pac != NULL: the two target groups fetch from the resident reference instead of packing raw bytes. */
__global__ __launch_bounds__(256) void bsw_pack_kernel(const uint8_t *__restrict__ raw, const bsw_dtask *__restrict__ tasks,
                                                       const bsw_rawoff *__restrict__ roff, const uint32_t bias, const uint32_t n,
                                                       const int rev_left, const uint8_t *__restrict__ pac, const int64_t l_pac,
                                                       const bsw_refx *__restrict__ refx, uint64_t *__restrict__ seq)
{
    const uint32_t g = blockIdx.x * 16u + (threadIdx.x >> 4);
    const int l16 = threadIdx.x & 15;
    const uint32_t ti = g >> 2;
    const int which = (int)(g & 3u);
    if (ti >= n) return;
    const bsw_dtask T = tasks[ti];
    if (pac && (which & 1)) {
        const bsw_refx X = refx[ti];
        const bool left = which == 1;
        const int tlen = left ? (T.lqlen ? T.ltlen : 0) : (T.rqlen ? T.rtlen : 0);
        const uint32_t woff = left ? T.lt_off : T.rt_off;
        const int nw = (tlen + 15) >> 4;
        for (int k = l16; k < nw; k += 16) seq[woff + (uint32_t)k] = fetch_word(pac, l_pac, left ? X.xl : X.xr, left ? -1 : 1, tlen, k);
        return;
    }
    const bsw_rawoff R = roff[ti];
    int len;
    uint32_t woff, boff;
    switch (which) {
    case 0: len = T.lqlen; woff = T.lq_off; boff = R.lq; break;
    case 1: len = T.lqlen ? T.ltlen : 0; woff = T.lt_off; boff = R.lt; break;
    case 2: len = T.rqlen; woff = T.rq_off; boff = R.rq; break;
    default: len = T.rqlen ? T.rtlen : 0; woff = T.rt_off; boff = R.rt; break;
    }
    boff -= bias;
    const int nw = (len + 15) >> 4;
    const bool rev = rev_left && which == 0;
    for (int k = l16; k < nw; k += 16) {
        /* forwards: bytes [off + 16k, +16); backwards: bytes (off - 16k - 16, off - 16k], then mirrored */
        const uintptr_t a = rev ? (uintptr_t)(raw + boff) - 16u * (uint32_t)k - 15u : (uintptr_t)(raw + boff) + 16u * (uint32_t)k;
        const uint32_t *q = (const uint32_t *)(a & ~(uintptr_t)3);
        const uint32_t sh = (uint32_t)(a & 3u);
        const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3], d4 = q[4];   /* the raw buffer has >= 32 bytes of slack on both sides */
        const uint32_t x0 = __builtin_amdgcn_alignbyte(d1, d0, sh), x1 = __builtin_amdgcn_alignbyte(d2, d1, sh);
        const uint32_t x2 = __builtin_amdgcn_alignbyte(d3, d2, sh), x3 = __builtin_amdgcn_alignbyte(d4, d3, sh);
        uint64_t lo = (uint64_t)x0 | ((uint64_t)x1 << 32), hi = (uint64_t)x2 | ((uint64_t)x3 << 32);
        if (rev) {
            const uint64_t t = __builtin_bswap64(hi);
            hi = __builtin_bswap64(lo);
            lo = t;
        }
        const int valid = len - 16 * k;                     /* bases of this word */
        if (valid < 16) {
            if (valid <= 8) { hi = 0; lo = valid == 8 ? lo : lo & ((1ull << (8 * valid)) - 1ull); }
            else hi &= (1ull << (8 * (valid - 8))) - 1ull;
        }
        lo = clamp_codes(lo);
        hi = clamp_codes(hi);
        seq[woff + (uint32_t)k] = squeeze8(lo) | (squeeze8(hi) << 32);
    }
}

/* ---- binning ---- */
/* does the packed sequence (16 bases per word, N = 4) hold an N?  (bases past `len` in the last word are not looked at) */
__device__ __forceinline__ int packed_has_n(const uint64_t *__restrict__ seq, const uint32_t off, const int len)
{
    const int nw = (len + 15) >> 4;
    uint64_t acc = 0;
    for (int k = 0; k < nw; ++k) {
        uint64_t v = seq[off + (uint32_t)k];
        if (k == nw - 1 && (len & 15)) v &= (1ull << (4 * (len & 15))) - 1ull;
        acc |= v;
    }
    return (acc & 0x4444444444444444ull) != 0;
}

struct seed_bins { int k0, k1, k2; };      /* class list, left-side bin, right-side bin (-1: none) */

__device__ __forceinline__ seed_bins seed_keys(const bsw_binparams &bp, const uint64_t *__restrict__ seq, const bsw_dtask &T)
{
    seed_bins s;
    s.k1 = s.k2 = -1;
    const int bits = bsw_seed_lane_bits(&bp, T.lqlen, T.rqlen, T.h0);
    if (!bits) {
        const int c = bsw_wave_class_of(&bp, T.lqlen > T.rqlen ? T.lqlen : T.rqlen);
        s.k0 = BSW_BIN_WAVE0 + (c < 0 ? 0 : c);     /* c < 0 cannot happen: the host rejects such seeds */
    } else {
        s.k0 = BSW_BIN_LANEALL;
        if (T.lqlen) s.k1 = BSW_BIN_L(bits == 16, packed_has_n(seq, T.lq_off, T.lqlen), bsw_h0_bucket(&bp, T.h0), T.lqlen);
        if (T.rqlen) s.k2 = BSW_BIN_R(bits == 16, packed_has_n(seq, T.rq_off, T.rqlen), T.rqlen);
    }
    return s;
}

__global__ __launch_bounds__(256) void bsw_bin_count(const bsw_binparams bp, const uint64_t *__restrict__ seq,
                                                     const bsw_dtask *__restrict__ tasks, const uint32_t n, uint32_t *__restrict__ bins)
{
    __shared__ uint32_t h[BSW_BIN_WAVE0];
    for (int b = threadIdx.x; b < BSW_BIN_WAVE0; b += 256) h[b] = 0;
    __syncthreads();
    for (uint32_t ti = blockIdx.x * 256u + threadIdx.x; ti < n; ti += gridDim.x * 256u) {
        const seed_bins s = seed_keys(bp, seq, tasks[ti]);
        if (s.k1 >= 0) atomicAdd(&h[s.k1], 1u);
        if (s.k2 >= 0) atomicAdd(&h[s.k2], 1u);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < BSW_BIN_WAVE0; b += 256)
        if (h[b]) atomicAdd(&bins[b], h[b]);
}

/* one block: per (side, lane class) turn the histograms into start offsets — the queries with an N, longest first, then the
 * ones without (a wave then holds equal-length queries and the most work starts first: a list that ENDED with the long N
 * queries left a few long waves running alone, 1 743 -> 1 339 GCUPS on the 250 bp workload); inside a query length the left
 * sides go by h0 bucket */
__global__ __launch_bounds__(256) void bsw_bin_scan(const bsw_binparams bp, uint32_t *__restrict__ bins)
{
    __shared__ uint32_t sc[256];
    const int t = threadIdx.x, q = 255 - t;
    for (int side = 0; side < 2; ++side)
        for (int c = 0; c < bp.n_lane; ++c) {
            uint32_t base = side ? bp.laneR_off[c] : bp.laneL_off[c];
            if ((side ? bp.laneR_off[c + 1] : bp.laneL_off[c + 1]) == base) continue;      /* (an empty list: wave-uniform, no barrier skipped by some) */
            const int bits = bp.lane_bits[c], b16 = bits == 16;
            const bool mine = bsw_side_lane_class(&bp, bits, q) == c;     /* (a folded class has no query length of its own) */
            for (int hn = 1; hn >= 0; --hn) {
                uint32_t part[BSW_H0_BUCKETS], v = 0;
                const int nb = side ? 1 : BSW_H0_BUCKETS;
#pragma unroll
                for (int hb = 0; hb < BSW_H0_BUCKETS; ++hb) {
                    part[hb] = (mine && hb < nb) ? bins[side ? BSW_BIN_R(b16, hn, q) : BSW_BIN_L(b16, hn, hb, q)] : 0u;
                    v += part[hb];
                }
                sc[t] = v;
                __syncthreads();
                for (int d = 1; d < 256; d <<= 1) {
                    const uint32_t add = t >= d ? sc[t - d] : 0u;
                    __syncthreads();
                    sc[t] += add;
                    __syncthreads();
                }
                uint32_t at = base + sc[t] - v;
                if (mine) {
#pragma unroll
                    for (int hb = 0; hb < BSW_H0_BUCKETS; ++hb)
                        if (hb < nb) { bins[side ? BSW_BIN_R(b16, hn, q) : BSW_BIN_L(b16, hn, hb, q)] = at; at += part[hb]; }
                }
                base += sc[255];
                __syncthreads();
            }
        }
    if (t < bp.n_wave) bins[BSW_BIN_WAVE0 + t] = bp.wave_start[t];
    if (t == 0) bins[BSW_BIN_LANEALL] = bp.lane_all_off;
}

/* BSW_SCATTER_TPT tasks per thread: the two passes over the cursor table (9 232 words) are shared by 1 024 tasks */
#define BSW_SCATTER_TPT 4
__global__ __launch_bounds__(256) void bsw_bin_scatter(const bsw_binparams bp, const uint64_t *__restrict__ seq,
                                                       const bsw_dtask *__restrict__ tasks, const uint32_t n,
                                                       uint32_t *__restrict__ bins, uint32_t *__restrict__ order)
{
    __shared__ uint32_t cnt[BSW_BIN_WORDS];               /* first the block's count per list, then — in place — its base in the list */
    for (int b = threadIdx.x; b < BSW_BIN_WORDS; b += 256) cnt[b] = 0;
    __syncthreads();
    seed_bins s[BSW_SCATTER_TPT];
    uint32_t r0[BSW_SCATTER_TPT], r1[BSW_SCATTER_TPT], r2[BSW_SCATTER_TPT];
#pragma unroll
    for (int k = 0; k < BSW_SCATTER_TPT; ++k) {
        const uint32_t ti = (blockIdx.x * BSW_SCATTER_TPT + k) * 256u + threadIdx.x;
        s[k].k0 = s[k].k1 = s[k].k2 = -1;
        r0[k] = r1[k] = r2[k] = 0;
        if (ti < n) {
            s[k] = seed_keys(bp, seq, tasks[ti]);
            r0[k] = atomicAdd(&cnt[s[k].k0], 1u);
            if (s[k].k1 >= 0) r1[k] = atomicAdd(&cnt[s[k].k1], 1u);
            if (s[k].k2 >= 0) r2[k] = atomicAdd(&cnt[s[k].k2], 1u);
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < BSW_BIN_WORDS; b += 256)
        if (cnt[b]) cnt[b] = atomicAdd(&bins[b], cnt[b]);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < BSW_SCATTER_TPT; ++k) {
        const uint32_t ti = (blockIdx.x * BSW_SCATTER_TPT + k) * 256u + threadIdx.x;
        if (ti < n) {
            order[cnt[s[k].k0] + r0[k]] = ti;
            if (s[k].k1 >= 0) order[cnt[s[k].k1] + r1[k]] = ti;
            if (s[k].k2 >= 0) order[cnt[s[k].k2] + r2[k]] = ti;
        }
    }
}

/* ---- reference wire format -> seq (F1).  One 16-lane group per sequence, as in bsw_pack_kernel.
 * `wire` holds whole 256 KiB task batches back to back; nib = nibble offset of the sequence's first base
 * counted from the start of `wire` (8 nibbles per 32-bit word, first base of a word in bits [31:28]). ---- */
__device__ __forceinline__ uint32_t rev_nibbles(uint32_t w)
{
    w = __builtin_bswap32(w);
    return ((w >> 4) & 0x0F0F0F0Fu) | ((w & 0x0F0F0F0Fu) << 4);
}

__global__ __launch_bounds__(256) void bsw_wire_pack_kernel(const uint32_t *__restrict__ wire, const bsw_dtask *__restrict__ tasks,
                                                            const bsw_wireoff *__restrict__ woffs, const uint32_t n,
                                                            uint64_t *__restrict__ seq)
{
    const uint32_t g = blockIdx.x * 16u + (threadIdx.x >> 4);
    const int l16 = threadIdx.x & 15;
    const uint32_t ti = g >> 2;
    const int which = (int)(g & 3u);
    if (ti >= n) return;
    const bsw_dtask T = tasks[ti];
    const bsw_wireoff W = woffs[ti];
    int len;
    uint32_t woff;
    uint32_t nib;                                           /* stream order: leftQ, rightQ, leftT, rightT */
    switch (which) {
    case 0: len = T.lqlen; woff = T.lq_off; nib = W.nib; break;
    case 1: len = T.lqlen ? T.ltlen : 0; woff = T.lt_off; nib = W.nib + W.lqlen + W.rqlen; break;
    case 2: len = T.rqlen; woff = T.rq_off; nib = W.nib + W.lqlen; break;
    default: len = T.rqlen ? T.rtlen : 0; woff = T.rt_off; nib = W.nib + W.lqlen + W.rqlen + W.ltlen; break;
    }
    const int nw = (len + 15) >> 4;
    for (int k = l16; k < nw; k += 16) {
        const uint32_t nb = nib + 16u * (uint32_t)k;
        const uint32_t *q = wire + (nb >> 3);
        const uint32_t sh = (uint32_t)(nb & 7u) * 4u;
        /* after rev_nibbles, base j of a wire word sits in bits [4j,4j+3] — the seq layout */
        const uint64_t a0 = rev_nibbles(q[0]), a1 = rev_nibbles(q[1]), a2 = rev_nibbles(q[2]);
        const uint64_t lo64 = a0 | (a1 << 32);
        uint64_t v = sh ? (lo64 >> sh) | (a2 << (64 - sh)) : lo64;
        const int valid = len - 16 * k;
        if (valid < 16) v &= (1ull << (4 * valid)) - 1ull;
        /* nibble codes > 4 -> 4 */
        const uint64_t big = ((v >> 3) | ((v >> 2) & ((v >> 1) | v))) & 0x1111111111111111ull;   /* 1 where nibble > 4 */
        v = (v & ~(big * 0xFull)) | (big << 2);
        seq[woff + (uint32_t)k] = v;
    }
}

/* ---- results -> the reference's 16 KiB result batches (F1): five words per task (rbb.v:59, proc_element.v:1662-1665,1190-1199),
 * the encoding of bsw_refbatch_encode_results, written where the task's batch and position say ---- */
__global__ __launch_bounds__(256) void bsw_wire_results_kernel(const bsw_result *__restrict__ out, const bsw_wireoff *__restrict__ woffs,
                                                               const uint32_t n, uint32_t *__restrict__ wout)
{
    const uint32_t ti = blockIdx.x * 256u + threadIdx.x;
    if (ti >= n) return;
    const uint4 a = *(const uint4 *)&out[ti], b = *((const uint4 *)&out[ti] + 1);      /* tag qb qe rb | re score truesc w */
    uint32_t *R = wout + woffs[ti].out_word;
    R[0] = a.x;
    R[1] = (a.z << 16) | (a.y & 0xffffu);
    R[2] = (b.x << 16) | (a.w & 0xffffu);
    R[3] = (b.z << 16) | (b.y & 0xffffu);
    R[4] = b.w;
}

/* ---- launchers ---- */
hipError_t launch_pack(const uint8_t *raw, const bsw_dtask *tasks, const bsw_rawoff *roff, uint32_t bias, uint32_t n, int rev_left,
                       const uint8_t *pac, int64_t l_pac, const bsw_refx *refx, uint64_t *seq, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(bsw_pack_kernel, dim3(n / 4u + (n % 4u ? 1u : 0u)), dim3(256), 0, s, raw, tasks, roff, bias, n, rev_left, pac, l_pac, refx, seq);
    return hipGetLastError();
}

hipError_t launch_wire_pack(const uint32_t *wire, const bsw_dtask *tasks, const bsw_wireoff *woffs, uint32_t n, uint64_t *seq, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    const uint32_t groups = n * 4u;
    hipLaunchKernelGGL(bsw_wire_pack_kernel, dim3((groups + 15u) / 16u), dim3(256), 0, s, wire, tasks, woffs, n, seq);
    return hipGetLastError();
}

hipError_t launch_wire_results(const bsw_result *out, const bsw_wireoff *woffs, uint32_t n, uint32_t *wout, size_t wout_words, hipStream_t s)
{
    hipError_t e = hipMemsetAsync(wout, 0, wout_words * sizeof(uint32_t), s);
    if (e != hipSuccess || n == 0) return e;
    hipLaunchKernelGGL(bsw_wire_results_kernel, dim3((n + 255u) / 256u), dim3(256), 0, s, out, woffs, n, wout);
    return hipGetLastError();
}

hipError_t launch_bin(const bsw_binparams &bp, const uint64_t *seq, const bsw_dtask *tasks, uint32_t n, uint32_t *bins, uint32_t *order, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(bins, 0, BSW_BIN_WORDS * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    const uint32_t blocks = (n + 255u) / 256u, sblocks = (n + 256u * BSW_SCATTER_TPT - 1u) / (256u * BSW_SCATTER_TPT);
    hipLaunchKernelGGL(bsw_bin_count, dim3(blocks > 1024u ? 1024u : blocks), dim3(256), 0, s, bp, seq, tasks, n, bins);
    hipLaunchKernelGGL(bsw_bin_scan, dim3(1), dim3(256), 0, s, bp, bins);
    hipLaunchKernelGGL(bsw_bin_scatter, dim3(sblocks), dim3(256), 0, s, bp, seq, tasks, n, bins, order);
    return hipGetLastError();
}

}  // namespace bsw
