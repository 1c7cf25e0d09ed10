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

/* One workgroup = 64 seeds = 256 sequences (leftQ, leftT, rightQ, rightT per seed); one thread per OUTPUT WORD.
 * Round 5's mapping — 16 lanes per sequence, lane k packing words k, k + 16, ... — left most lanes idle (a PE query is 3 - 9
 * words, a target 5 - 14): 1.58 ms per 4 Mi PE seeds, 1.7 TB/s.  Here thread t of the group first describes sequence t (length,
 * word offset, byte offset), the group prefix-sums the word counts, and then the threads walk the group's words 256 at a
 * time, each finding its sequence by a binary search over the prefix sums in LDS (8 probes): every lane packs a word, a
 * wavefront's stores are consecutive words of `seq`, its loads consecutive 16-byte pieces of `raw`.
 * rev_left: the left query is read BACKWARDS from its offset (base k = raw[off - k]): a read DMA'd as it is holds
 * query[0..qbeg) forwards, mem_chain2aln extends it reversed (the host's reversal loop, done here for free).
 * pac != NULL: the two target sequences are fetched from the resident reference instead of packed from raw bytes.
 * nflag != NULL: nflag[seed] = (left query holds an N) | (right query holds an N) << 1 — what the binning needs to know about
 * the bases, so that bsw_bin_count does not have to read `seq` again. */
#define BSW_PACK_SEEDS 64
__global__ __launch_bounds__(256) void bsw_pack_kernel(const uint8_t *__restrict__ raw, const bsw_dtask *__restrict__ tasks,
                                                       const bsw_rawoff *__restrict__ roff, const uint32_t bias, const uint32_t n,
                                                       const int rev_left, const uint8_t *__restrict__ pac, const int64_t l_pac,
                                                       const bsw_refx *__restrict__ refx, uint64_t *__restrict__ seq, uint8_t *__restrict__ nflag)
{
    __shared__ uint32_t start[257];               /* start[s] = words of the group's sequences before s */
    __shared__ uint32_t s_woff[256], s_boff[256], s_hasn[256];
    __shared__ int s_len[256];
    __shared__ uint32_t wsum[4];
    const int sq = threadIdx.x, which = sq & 3;
    const uint32_t ti = blockIdx.x * (uint32_t)BSW_PACK_SEEDS + (uint32_t)(sq >> 2);
    int len = 0;
    uint32_t woff = 0, boff = 0;
    if (ti < n) {
        const bsw_dtask T = tasks[ti];
        const bool from_pac = pac && (which & 1);
        bsw_rawoff R = {0, 0, 0, 0};
        if (!from_pac) R = roff[ti];
        switch (which) {
        case 0: len = T.lqlen; woff = T.lq_off; boff = R.lq; break;
        case 1: len = T.lqlen ? T.ltlen : 0; woff = T.lt_off; boff = R.lt; break;
        case 2: len = T.rqlen; woff = T.rq_off; boff = R.rq; break;
        default: len = T.rqlen ? T.rtlen : 0; woff = T.rt_off; boff = R.rt; break;
        }
        boff -= bias;
    }
    const uint32_t nw = (uint32_t)(len + 15) >> 4;
    /* inclusive scan of nw inside the wavefront, then across the four wavefronts */
    uint32_t inc = nw;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)inc, d, 64);
        if ((sq & 63) >= d) inc += up;
    }
    if ((sq & 63) == 63) wsum[sq >> 6] = inc;
    s_woff[sq] = woff; s_boff[sq] = boff; s_len[sq] = len; s_hasn[sq] = 0;
    __syncthreads();
    uint32_t before = 0;
    for (int k = 0; k < (sq >> 6); ++k) before += wsum[k];
    start[sq] = before + inc - nw;
    if (sq == 255) start[256] = before + inc;
    __syncthreads();
    const uint32_t total = start[256];
    for (uint32_t w = (uint32_t)sq; w < total; w += 256u) {
        int lo = 0;                                            /* the last sequence that starts at or before word w */
#pragma unroll
        for (int step = 128; step >= 1; step >>= 1)
            if (start[lo + step] <= w) lo += step;
        const int k = (int)(w - start[lo]);
        const int wh = lo & 3, L = s_len[lo];
        const uint32_t tsk = blockIdx.x * (uint32_t)BSW_PACK_SEEDS + (uint32_t)(lo >> 2);
        uint64_t v;
        if (pac && (wh & 1)) {
            const bsw_refx X = refx[tsk];
            const bool left = wh == 1;
            v = fetch_word(pac, l_pac, left ? X.xl : X.xr, left ? -1 : 1, L, k);
        } else {
            const bool rev = rev_left && wh == 0;
            const uint8_t *base = raw + s_boff[lo];
            /* forwards: bytes [off + 16k, +16); backwards: bytes (off - 16k - 16, off - 16k], then mirrored */
            const uintptr_t a = rev ? (uintptr_t)base - 16u * (uint32_t)k - 15u : (uintptr_t)base + 16u * (uint32_t)k;
            const uint32_t *q = (const uint32_t *)(a & ~(uintptr_t)3);
            const uint32_t sh = (uint32_t)(a & 3u);
            const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3], d4 = q[4];   /* the raw buffer has >= 32 bytes of slack on both sides */
            const uint32_t x0 = __builtin_amdgcn_alignbyte(d1, d0, sh), x1 = __builtin_amdgcn_alignbyte(d2, d1, sh);
            const uint32_t x2 = __builtin_amdgcn_alignbyte(d3, d2, sh), x3 = __builtin_amdgcn_alignbyte(d4, d3, sh);
            uint64_t lo64 = (uint64_t)x0 | ((uint64_t)x1 << 32), hi64 = (uint64_t)x2 | ((uint64_t)x3 << 32);
            if (rev) {
                const uint64_t t = __builtin_bswap64(hi64);
                hi64 = __builtin_bswap64(lo64);
                lo64 = t;
            }
            const int valid = L - 16 * k;                     /* bases of this word */
            if (valid < 16) {
                if (valid <= 8) { hi64 = 0; lo64 = valid == 8 ? lo64 : lo64 & ((1ull << (8 * valid)) - 1ull); }
                else hi64 &= (1ull << (8 * (valid - 8))) - 1ull;
            }
            lo64 = clamp_codes(lo64);
            hi64 = clamp_codes(hi64);
            v = squeeze8(lo64) | (squeeze8(hi64) << 32);
        }
        seq[s_woff[lo] + (uint32_t)k] = v;
        if (!(wh & 1) && (v & 0x4444444444444444ull)) s_hasn[lo] = 1u;      /* (a query word with an N: rare; any writer writes the same 1) */
    }
    if (nflag) {
        __syncthreads();
        if (which == 0 && ti < n) nflag[ti] = (uint8_t)((s_hasn[sq] ? 1u : 0u) | (s_hasn[sq + 2] ? 2u : 0u));
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

/* nflag: the pack kernel's per-seed N bits (NULL: input that came packed already — the queries' words are read here) */
__device__ __forceinline__ seed_bins seed_keys(const bsw_binparams &bp, const uint64_t *__restrict__ seq, const uint8_t *__restrict__ nflag,
                                               const uint32_t ti, const bsw_dtask &T)
{
    seed_bins s;
    s.k1 = s.k2 = -1;
    const int bits = bsw_seed_lane_bits(&bp, T.lqlen, T.rqlen, T.h0);
    if (!bits) {
        const int c = bsw_wave_class_of(&bp, T.lqlen > T.rqlen ? T.lqlen : T.rqlen);
        s.k0 = BSW_BIN_WAVE0 + (c < 0 ? 0 : c);     /* c < 0 cannot happen: the host rejects such seeds */
    } else {
        s.k0 = BSW_BIN_LANEALL;
        const int nf = nflag ? (int)nflag[ti] : -1;
        if (bp.nsplit && bits == 8 && (nf >= 0 ? (nf & 3) != 0 : (packed_has_n(seq, T.lq_off, T.lqlen) | packed_has_n(seq, T.rq_off, T.rqlen)) != 0)) {
            s.k0 = BSW_BIN_NLIST;               /* (bsw_binparams.nsplit) */
            return s;
        }
        const bool fz = bp.fused && bits == 8;       /* both sides of an 8-bit seed in one launch: on the left lists whatever its left side, on no right list */
        /* (a fused launch runs both queries of a seed.  The group kernel's — fused == 1 — puts the seeds with an N in EITHER query in
         * front, so that the other wavefronts hold no N in either half: 32 k PE seeds with Ns 1.10 -> 1.00 ms.  The lane kernel's
         * — fused == 2 — keeps the left query's N alone as the key: a wavefront with an N in nearly every block of BOTH halves is
         * the launch's longest, 131 k seeds 2.27 -> 3.0 ms; tools/diag/ab_nfirst.py) */
        const int nmask = bp.fused == 1 ? 3 : 1;
        if (fz) s.k1 = BSW_BIN_L(0, nf >= 0 ? ((nf & nmask) != 0) : (packed_has_n(seq, T.lq_off, T.lqlen) | (nmask == 3 ? packed_has_n(seq, T.rq_off, T.rqlen) : 0)), bsw_h0_bucket(&bp, T.h0), T.lqlen);
        else if (T.lqlen) s.k1 = BSW_BIN_L(bits == 16, nf >= 0 ? (nf & 1) : packed_has_n(seq, T.lq_off, T.lqlen), bsw_h0_bucket(&bp, T.h0), T.lqlen);
        if (T.rqlen && !fz) s.k2 = BSW_BIN_R(bits == 16, nf >= 0 ? ((nf >> 1) & 1) : packed_has_n(seq, T.rq_off, T.rqlen), T.rqlen);
    }
    return s;
}

/* the three list indices of a seed in one word (0xffff: none): bsw_bin_count works them out once, bsw_bin_scatter reads 8 bytes
 * per seed instead of the task record and the query words again */
__device__ __forceinline__ uint64_t keys_pack(const seed_bins &s)
{
    return (uint64_t)(uint32_t)s.k0 | ((uint64_t)(uint16_t)s.k1 << 16) | ((uint64_t)(uint16_t)s.k2 << 32);
}
static_assert(BSW_BIN_WORDS < 0xffff, "bin indices are kept in 16 bits");

__global__ __launch_bounds__(256) void bsw_bin_count(const bsw_binparams bp, const uint64_t *__restrict__ seq, const uint8_t *__restrict__ nflag,
                                                     const bsw_dtask *__restrict__ tasks, const uint32_t n, uint32_t *__restrict__ bins,
                                                     uint64_t *__restrict__ keys)
{
    __shared__ uint32_t h[BSW_BIN_WAVE0];
    for (int b = threadIdx.x; b < BSW_BIN_WAVE0; b += 256) h[b] = 0;
    __syncthreads();
    for (uint32_t ti = blockIdx.x * 256u + threadIdx.x; ti < n; ti += gridDim.x * 256u) {
        const seed_bins s = seed_keys(bp, seq, nflag, ti, tasks[ti]);
        keys[ti] = keys_pack(s);
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
    if (t == 0) { bins[BSW_BIN_LANEALL] = bp.lane_all_off; bins[BSW_BIN_NLIST] = bp.nlist_off; }
}

/* bsw_binparams.nsplit: the N list's length where the general kernel's launch reads it */
__global__ void bsw_nlist_count(const uint32_t *__restrict__ bins, const uint32_t nlist_off, uint32_t *__restrict__ dst)
{
    *dst = bins[BSW_BIN_NLIST] - nlist_off;
}

/* BSW_SCATTER_TPT tasks per thread: the two passes over the cursor table (9 232 words) are shared by 1 024 tasks.  The class
 * list of a seed (k0) is one of a handful of values — nearly always the list of all lane seeds — so its rank inside the
 * workgroup is taken per WAVEFRONT (one LDS atomic per distinct value and wavefront, ranks from the lane mask) instead of
 * 1 024 atomics on one LDS word. */
#define BSW_SCATTER_TPT 4
__global__ __launch_bounds__(256) void bsw_bin_scatter(const uint64_t *__restrict__ keys, const uint32_t n,
                                                       uint32_t *__restrict__ bins, uint32_t *__restrict__ order)
{
    __shared__ uint32_t cnt[BSW_BIN_WORDS];               /* first the block's count per list, then — in place — its base in the list */
    for (int b = threadIdx.x; b < BSW_BIN_WORDS; b += 256) cnt[b] = 0;
    __syncthreads();
    int k0[BSW_SCATTER_TPT], k1[BSW_SCATTER_TPT], k2[BSW_SCATTER_TPT];
    uint32_t r0[BSW_SCATTER_TPT], r1[BSW_SCATTER_TPT], r2[BSW_SCATTER_TPT];
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < BSW_SCATTER_TPT; ++k) {
        const uint32_t ti = (blockIdx.x * BSW_SCATTER_TPT + k) * 256u + threadIdx.x;
        k0[k] = k1[k] = k2[k] = -1;
        r0[k] = r1[k] = r2[k] = 0;
        if (ti < n) {
            const uint64_t K = keys[ti];
            k0[k] = (int)(K & 0xffffu);
            k1[k] = (int)((K >> 16) & 0xffffu); if (k1[k] == 0xffff) k1[k] = -1;
            k2[k] = (int)((K >> 32) & 0xffffu); if (k2[k] == 0xffff) k2[k] = -1;
        }
        /* rank in the class list: the lanes of the wavefront that share a value take consecutive places */
        uint64_t todo = __ballot(k0[k] >= 0);
        while (todo) {
            const int lead = __builtin_ctzll(todo);
            const int val = __shfl(k0[k], lead, 64);
            const uint64_t same = __ballot(k0[k] == val) & todo;
            uint32_t basev = 0;
            if (lane == lead) basev = atomicAdd(&cnt[val], (uint32_t)__builtin_popcountll(same));
            basev = (uint32_t)__shfl((int)basev, lead, 64);
            if (k0[k] == val) r0[k] = basev + (uint32_t)__builtin_popcountll(same & ((1ull << lane) - 1ull));
            todo &= ~same;
        }
        if (k1[k] >= 0) r1[k] = atomicAdd(&cnt[k1[k]], 1u);
        if (k2[k] >= 0) r2[k] = atomicAdd(&cnt[k2[k]], 1u);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < BSW_BIN_WORDS; b += 256)
        if (cnt[b]) cnt[b] = atomicAdd(&bins[b], cnt[b]);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < BSW_SCATTER_TPT; ++k) {
        const uint32_t ti = (blockIdx.x * BSW_SCATTER_TPT + k) * 256u + threadIdx.x;
        if (ti < n) {
            order[cnt[k0[k]] + r0[k]] = ti;
            if (k1[k] >= 0) order[cnt[k1[k]] + r1[k]] = ti;
            if (k2[k] >= 0) order[cnt[k2[k]] + r2[k]] = ti;
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

/* ---- word offsets made absolute (the host's pass over a chunk runs on several threads, each counting from 0) ---- */
__global__ __launch_bounds__(256) void bsw_rebase_kernel(bsw_dtask *__restrict__ tasks, const bsw_rawoff *__restrict__ roff, const uint32_t n, const bsw_rebase rb)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    uint32_t *T = (uint32_t *)&tasks[i];                       /* lq_off, lt_off, rq_off, rt_off are its first four words */
    if (rb.use_ro) {
        const bsw_rawoff R = roff[i];
        T[0] = R.lq + rb.delta; T[1] = R.lt + rb.delta; T[2] = R.rq + rb.delta; T[3] = R.rt + rb.delta;
    } else {
        uint32_t k = i / rb.per;
        if (k >= rb.nr) k = rb.nr - 1u;
        const uint32_t b = rb.base[k];
        T[0] += b; T[1] += b; T[2] += b; T[3] += b;
    }
}

/* ---- launchers ---- */
hipError_t launch_rebase(bsw_dtask *tasks, const bsw_rawoff *roff, uint32_t n, const bsw_rebase &rb, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(bsw_rebase_kernel, dim3((n + 255u) / 256u), dim3(256), 0, s, tasks, roff, n, rb);
    return hipGetLastError();
}

hipError_t launch_pack(const uint8_t *raw, const bsw_dtask *tasks, const bsw_rawoff *roff, uint32_t bias, uint32_t n, int rev_left,
                       const uint8_t *pac, int64_t l_pac, const bsw_refx *refx, uint64_t *seq, uint8_t *nflag, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(bsw_pack_kernel, dim3((n + BSW_PACK_SEEDS - 1u) / BSW_PACK_SEEDS), dim3(256), 0, s, raw, tasks, roff, bias, n, rev_left, pac, l_pac, refx, seq, nflag);
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

hipError_t launch_bin(const bsw_binparams &bp, const uint64_t *seq, const uint8_t *nflag, const bsw_dtask *tasks, uint32_t n, uint32_t *bins,
                      uint64_t *keys, uint32_t *order, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(bins, 0, BSW_BIN_WORDS * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    if (bp.nsplit && bp.fill_len && (e = hipMemsetAsync(order + bp.fill_off, 0xff, (size_t)bp.fill_len * sizeof(uint32_t), s)) != hipSuccess) return e;
    const uint32_t blocks = (n + 255u) / 256u, sblocks = (n + 256u * BSW_SCATTER_TPT - 1u) / (256u * BSW_SCATTER_TPT);
    hipLaunchKernelGGL(bsw_bin_count, dim3(blocks > 2048u ? 2048u : blocks), dim3(256), 0, s, bp, seq, nflag, tasks, n, bins, keys);
    hipLaunchKernelGGL(bsw_bin_scan, dim3(1), dim3(256), 0, s, bp, bins);
    hipLaunchKernelGGL(bsw_bin_scatter, dim3(sblocks), dim3(256), 0, s, keys, n, bins, order);
    if (bp.nsplit) hipLaunchKernelGGL(bsw_nlist_count, dim3(1), dim3(1), 0, s, bins, bp.nlist_off, order + bp.nlist_cnt_at);
    return hipGetLastError();
}

}  // namespace bsw
