/*
 * bsw_device.h — device-side data layout shared by the host batch manager and the
 * HIP kernels (internal; the public C ABI is include/bwa_sw_mi355.h).
 *
 * HBM layout of one device batch
 *   raw   : uint8[]    the caller's byte-per-base sequences exactly as they arrived over PCIe
 *                      (one DMA of a registered arena, or a gather into pinned staging);
 *                      consumed once by bsw_pack_kernel, never read by the DP kernels
 *   seq   : uint64[]   all sequences, 4 bits per base, 16 bases per uint64, base k of
 *                      a word in bits [4k,4k+3]; every sequence starts on a word
 *                      boundary; codes 0..3 = ACGT, 4 = N (anything >4 is stored as 4)
 *   tasks : bsw_dtask[] one record per seed (the RTL's 8-word header H0..H7,
 *                      sw_pe_array_proc_element.v:807-933, widened past its 8-bit limits)
 *   order : uint32[]   launch order -> task index (bins built on the device by bsw_bin_*)
 *   out   : bsw_result[] indexed by task index (task order, not completion order —
 *                      the RTL's fill_resulBuf emits completion order and needs the tag)
 */
#ifndef BSW_DEVICE_H
#define BSW_DEVICE_H

#include <stdint.h>
#include "../../include/bwa_sw_mi355.h"

#if defined(__HIPCC__) || defined(__HIP__)
#define BSW_HD __host__ __device__ inline
#else
#define BSW_HD static inline
#endif

typedef struct bsw_dtask {
    uint32_t lq_off, lt_off, rq_off, rt_off;   /* word offsets into seq              */
    uint16_t lqlen, rqlen, ltlen, rtlen;
    uint16_t wlim_l, wlim_r;                   /* min(max_ins,max_del) per side (H5/H6) */
    int32_t  h0, init_score, qbeg;
    uint32_t tag;
} bsw_dtask;                                   /* 44 bytes */

/* where the four byte-per-base sequences of a seed start inside `raw` (bytes) */
typedef struct bsw_rawoff {
    uint32_t lq, lt, rq, rt;
} bsw_rawoff;

/* what bsw_rebase_kernel does to the task records of a chunk: the host writes word offsets relative to the RANGE of seeds a
 * thread walked (bsw_batch.hip: prepare_chunk_t), the device adds the range's base; packed input DMA'd as it lies takes its
 * offsets from rawoff = (host pointer >> 3) mod 2^32 plus delta = -(arena base >> 3) */
#define BSW_REBASE_MAX 16
typedef struct bsw_rebase {
    uint32_t per, nr;                          /* seeds per range, ranges */
    uint32_t use_ro, delta;
    uint32_t base[BSW_REBASE_MAX];
} bsw_rebase;

typedef struct bsw_dparams {
    int8_t  mat[25];
    int8_t  pad[3];
    int32_t o_del, e_del, o_ins, e_ins;
    int32_t w, pen_clip5, pen_clip3, zdrop, max_band_try;
} bsw_dparams;

/* where a seed's two targets start in the device-resident 2-bit reference, in bwa's [0, 2*l_pac) coordinates
 * (bsw_pack_kernel fetches them: ltlen bases downwards from xl, rtlen bases upwards from xr) */
typedef struct bsw_refx {
    int64_t xl;           /* seed.rbeg - 1            */
    int64_t xr;           /* seed.rbeg + seed.len     */
} bsw_refx;

/* one banded global alignment (bsw_global_kernel.hip) */
typedef struct bsw_gdtask {
    uint32_t q_off, t_off;    /* word offsets into seq */
    int32_t  qlen, tlen, w;
    uint32_t pad;
    uint64_t z_off;           /* byte offset of this alignment's backtrack matrix */
} bsw_gdtask;

/* one local alignment (bsw_align_kernel.hip; SURVEY.md §8f F4: bwa ksw_align2) */
typedef struct bsw_adtask {
    uint32_t q_off, t_off;    /* word offsets into seq */
    int32_t  qlen, tlen, xtra;
    uint32_t pad;
    uint64_t b_off;           /* first entry of this alignment's slice of the sub-optimal list scratch */
} bsw_adtask;

#define BSW_KEY_BITS 10                        /* column index bits in the arg-max key */

/* ---- binning: which kernel class a seed goes to.  The host counts seeds per class with these
 * functions (launch sizes), the device sorts with the same functions (bsw_stage_kernel.hip), so the
 * two can never disagree. ---- */
#define BSW_MAX_WAVE_CLASSES 8
#define BSW_MAX_LANE_CLASSES 4
#define BSW_LANE_QBINS       256               /* a lane class holds at most 256 eh[] columns */

typedef struct bsw_binparams {
    int32_t a;                                 /* match score (score range test)            */
    int32_t b;                                 /* mismatch penalty (>= 0): the two-seeds-per-lane kernel forms H + a + b
                                                  in 8 bits before it subtracts b                                        */
    int32_t lane_on;                           /* 0: every seed goes to the wave-per-task classes */
    int32_t n_wave, n_lane;
    int32_t wave_cols[BSW_MAX_WAVE_CLASSES];
    int32_t lane_cols[BSW_MAX_LANE_CLASSES];
    int32_t lane_bits[BSW_MAX_LANE_CLASSES];   /* 8 / 16: width of h and e in the lane class */
    int32_t cols8, cols16;                     /* widest lane class per value width          */
    /* offsets into `order` of every list the device fills */
    uint32_t wave_start[BSW_MAX_WAVE_CLASSES + 1];
    uint32_t lane_all_off;
    uint32_t laneL_off[BSW_MAX_LANE_CLASSES + 1], laneR_off[BSW_MAX_LANE_CLASSES + 1];
    /* second sort key of the LEFT sides inside a query length: the seed's h0 in BSW_H0_BUCKETS buckets over the chunk's range
     * [hb_lo, ...], bucket = ((h0 - hb_lo) * hb_mul) >> 16 (hb_mul = 0: one bucket) */
    int32_t hb_lo, hb_mul;
    /* != 0: a mid-sized chunk whose two sides run in ONE launch (1: bsw_lane2g_kernel, 2: bsw_lane2_kernel; side 2): every 8-bit lane
     * seed is on the LEFT lists, by its left query length (0: it has no left side), and no 8-bit right list is filled */
    int32_t fused;
    /* 1: a chunk that does not fill the machine — its launches last as long as their slowest wavefront, and a wavefront of the
     * two-seeds-per-lane kernels that holds queries with an N runs 1.6 - 2x as long as the others (its N bodies are cold code,
     * meant for a stray block).  The 8-bit lane seeds with an N in either query then go on a list of their own, order[nlist_off ..],
     * for the general kernel (one wavefront per seed, N or not), and sit on no lane list; the lists' unused tails hold
     * 0xffffffff (BSW_ORDER_NONE).  The list's length is left in order[nlist_cnt_at]. */
    int32_t nsplit;
    uint32_t nlist_off, nlist_cap, nlist_cnt_at, fill_off, fill_len;
} bsw_binparams;
#define BSW_ORDER_NONE 0xffffffffu

/* The lanes of a wave walk their rows in lockstep over the UNION of their [beg, end) ranges, and h0 sets how fast a seed's
 * range opens (end ~ 2 i + h0 - o) and where its beg runs later: seeds of one query length but h0 = 19 .. 60 spread their range ends
 * over five 8-column blocks.  (A right side's h0 is the left side's score ~ h0 + lqlen = read length - rqlen: the same for
 * every seed of its query-length bin already.)  PE mixed bins, 1 M seeds, handed over sorted by h0 — the bins fill in arrival order —
 * ran 2 480 -> 2 542 GCUPS (profiles/r5/h0_second_sort_key.txt). */
#define BSW_H0_BUCKETS 8
BSW_HD int bsw_h0_bucket(const bsw_binparams *bp, int h0)
{
    int64_t d = (int64_t)h0 - bp->hb_lo;
    if (d < 0) d = 0;
    d = (d * bp->hb_mul) >> 16;
    return d >= BSW_H0_BUCKETS ? BSW_H0_BUCKETS - 1 : (int)d;
}
/* hb_lo / hb_mul for a chunk whose left sides have h0 in [lo, hi] */
BSW_HD void bsw_set_h0_buckets(bsw_binparams *bp, int lo, int hi)
{
    bp->hb_lo = lo;
    bp->hb_mul = hi > lo ? (int32_t)(((int64_t)BSW_H0_BUCKETS << 16) / ((int64_t)hi - lo + 1)) : 0;
}

/* value width of the lane kernel a seed may use: 0 = wave-per-task kernel */
BSW_HD int bsw_seed_lane_bits(const bsw_binparams *bp, int lqlen, int rqlen, int h0)
{
    if (!bp->lane_on) return 0;
    const int qm = lqlen > rqlen ? lqlen : rqlen;
    if (qm == 0) return 0;                     /* no side at all: nothing a lane launch would finish (the general kernel writes the neutral record) */
    const int64_t top = (int64_t)h0 + (int64_t)(lqlen + rqlen) * bp->a;   /* no H can exceed this */
    if (top + bp->b <= 255 && qm + 1 <= bp->cols8) return 8;
    if (top < 65000 && qm + 1 <= bp->cols16) return 16;
    return 0;
}
/* lane class of one side (query length q) of a seed with value width `bits`; -1: none */
BSW_HD int bsw_side_lane_class(const bsw_binparams *bp, int bits, int q)
{
    for (int c = 0; c < bp->n_lane; ++c)
        if (bp->lane_bits[c] == bits && q + 1 <= bp->lane_cols[c]) return c;
    return -1;
}
BSW_HD int bsw_wave_class_of(const bsw_binparams *bp, int qmax)
{
    for (int c = 0; c < bp->n_wave; ++c)
        if (qmax + 1 <= bp->wave_cols[c]) return c;
    return -1;
}

/* ---- the pair-level decision (P1 exit test, P2 clip-vs-extend, P3: sw_pe_array_proc_element.v:1593-1685) for a seed whose
 * sides came from the lane kernels.  ONE function for its two callers: bsw_pair_finalize (a launch of its own) and the
 * epilogue of the two-seeds-per-lane kernels (round 5: the launch that computes a seed's LAST side finishes the seed itself —
 * one launch and a 96-byte read-modify-write per seed less).  L / R: the sides' K9 records (ignored for a side that does not
 * exist).  A seed whose first band try was not final goes to the redo list instead of getting a record. ---- */
typedef struct bsw_fin {
    uint32_t *redo, *redo_cnt;                 /* the chunk's redo list and its length (device words) */
    bsw_pair *pairs;                           /* BSW_RESULT_PAIR: the dense 32-byte records; else NULL */
    int on;                                    /* 0: the lane kernels only store their side (bsw_pair_finalize follows) */
    int group;                                 /* 1: the chunk is mid-sized — its 8-bit lane launches run bsw_lane2g_kernel (a seed pair
                                                  per group of eight lanes, 16 seeds per wavefront) instead of the 128-seed kernels;
                                                  2: and both sides of a seed in one launch (bsw_binparams.fused) */
} bsw_fin;

#if defined(__HIPCC__) || defined(__HIP__)
__device__ __forceinline__ void bsw_pair_decide(const bsw_dparams &P, const bsw_dtask &T, const uint32_t ti, bsw_ext L, bsw_ext R,
                                                bsw_result *__restrict__ out, uint32_t *__restrict__ redo, uint32_t *__restrict__ redo_cnt,
                                                bsw_pair *__restrict__ pairs)
{
    const int thr = (P.w >> 1) + (P.w >> 2);
    bool retry = false;
    int score = T.init_score, truesc, qb, rb, qe, re;
    bsw_ext Z;
    Z.score = 0; Z.qle = Z.tle = Z.gtle = 0; Z.gscore = 0; Z.max_off = 0; Z.aw = P.w; Z.cells = 0;
    if (T.lqlen > 0) {
        if (P.max_band_try > 1 && !(L.score == score || L.max_off < thr)) retry = true;
        score = L.score;
        if (L.gscore <= 0 || L.gscore <= score - P.pen_clip5) { qb = T.qbeg - L.qle; rb = -L.tle; truesc = score; }
        else { qb = 0; rb = -L.gtle; truesc = L.gscore; }
    } else {
        score = truesc = T.h0; qb = 0; rb = 0;
        L = Z;
    }
    const int sc0 = score;
    if (T.rqlen > 0) {
        if (P.max_band_try > 1 && !(R.score == sc0 || R.max_off < thr)) retry = true;
        score = R.score;
        if (R.gscore <= 0 || R.gscore <= score - P.pen_clip3) { qe = R.qle; re = R.tle; truesc += score - sc0; }
        else { qe = T.rqlen; re = R.gtle; truesc += R.gscore - sc0; }
    } else {
        qe = 0; re = 0;
        R = Z;
    }
    if (retry) {
        redo[atomicAdd(redo_cnt, 1u)] = ti;
        return;
    }
    if (pairs) {
        /* BSW_RESULT_PAIR: the RTL's 5-word record alone (sw_pe_array_proc_element.v:1662-1665), 32 bytes into the dense
         * array that crosses PCIe; the per-side records stay where the lane kernels left them */
        bsw_pair pr;
        pr.tag = T.tag; pr.qb = qb; pr.qe = qe; pr.rb = rb; pr.re = re; pr.score = score; pr.truesc = truesc; pr.w = P.w;
        pairs[ti] = pr;
    } else {
        bsw_result r;
        r.tag = T.tag; r.qb = qb; r.qe = qe; r.rb = rb; r.re = re; r.score = score; r.truesc = truesc; r.w = P.w;
        r.left = L; r.right = R;
        out[ti] = r;
    }
}
#endif

#endif
