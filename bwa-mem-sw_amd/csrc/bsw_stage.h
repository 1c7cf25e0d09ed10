/* bsw_stage.h — internal: launchers of the kernels in bsw_stage_kernel.hip / bsw_*_kernel.hip */
#ifndef BSW_STAGE_H
#define BSW_STAGE_H

#include <hip/hip_runtime.h>
#include "bsw_device.h"

/* layout of the `bins` scratch array (uint32): histogram -> cursor per
 *   left sides : (16-bit class?, query holds an N, h0 bucket, query length)
 *   right sides: (16-bit class?, query holds an N, query length)
 * (the lane class of a side follows from its width and query length: bsw_side_lane_class), then one cursor per wave class list,
 * then the cursor of the list of all lane seeds */
#define BSW_BIN_L(b16, has_n, hb, q) (((((b16) * 2 + (has_n)) * BSW_H0_BUCKETS + (hb)) * BSW_LANE_QBINS) + (q))
#define BSW_BIN_R0      (2 * 2 * BSW_H0_BUCKETS * BSW_LANE_QBINS)
#define BSW_BIN_R(b16, has_n, q) (BSW_BIN_R0 + ((b16) * 2 + (has_n)) * BSW_LANE_QBINS + (q))
#define BSW_BIN_WAVE0   (BSW_BIN_R0 + 2 * 2 * BSW_LANE_QBINS)
#define BSW_BIN_LANEALL (BSW_BIN_WAVE0 + BSW_MAX_WAVE_CLASSES)
#define BSW_BIN_NLIST   (BSW_BIN_LANEALL + 1)     /* bsw_binparams.nsplit: the 8-bit lane seeds with an N in a query (general kernel) */
#define BSW_BIN_WORDS   (BSW_BIN_LANEALL + 8)

/* where a task's nibble stream starts inside the uploaded wire batches (256 batches x 65 536 words x 8 nibbles = 2^27), where
 * its 5-word record goes inside the group's result batches, and the four lengths in stream order */
typedef struct bsw_wireoff {
    uint32_t nib, out_word;
    uint16_t lqlen, rqlen, ltlen, rtlen;
} bsw_wireoff;

namespace bsw {
int wave_class_count();
int wave_class_cols(int cls);
hipError_t launch_wave(int cls, int variant, const bsw_dparams &P, const uint64_t *seq, const bsw_dtask *tasks,
                       const uint32_t *order, uint32_t n, const uint32_t *n_dev, uint32_t *next_slot, bsw_result *out, hipStream_t s);
int lane_class_count();
int lane_class_cols(int cls);
int lane_class_bits(int cls);
bool lane_class_signals_tail(int cls);    /* its launch raises launch_lane's tail_flag when the last workgroup starts */
/* tail_flag != NULL: *tail_flag (a device word, zero before the launch) reaches *tail_target when every workgroup of the
 * launch has started — or, from a kernel that cannot say so, when the launch is done */
/* fin != NULL && fin->on: the launch that computes a seed's last side also takes the pair-level decision and writes the
 * whole record / the redo list (every class of the chunk must say lane_class_finishes) */
hipError_t launch_lane(int cls, int variant, const bsw_dparams &P, int side, const uint64_t *seq, const bsw_dtask *tasks,
                       const uint32_t *order, uint32_t n, bsw_result *out, hipStream_t s, uint32_t *tail_flag = nullptr, uint32_t *tail_target = nullptr,
                       const bsw_fin *fin = nullptr);
bool lane_class_finishes(int cls, const bsw_dparams &P, int variant);
/* pairs != NULL (BSW_RESULT_PAIR): the pair-level record goes to pairs[task], 32 bytes, instead of back into out[task] */
hipError_t launch_wait_count(const uint32_t *flag, uint32_t target, uint32_t *expired, hipStream_t s);    /* one sleeping wave until *flag >= target or 20 ms */
hipError_t launch_finalize(const bsw_dparams &P, const bsw_dtask *tasks, const uint32_t *order, uint32_t n,
                           bsw_result *out, uint32_t *redo, uint32_t *redo_cnt, bsw_pair *pairs, hipStream_t s);
hipError_t launch_pairs_from_results(const uint32_t *order, uint32_t n, const uint32_t *n_dev, const bsw_result *out, bsw_pair *pairs, hipStream_t s);
/* raw byte of a sequence = raw[roff - bias] (uint32 arithmetic).  pac != NULL: the targets are not in raw, they are
 * fetched from the resident reference at refx[] */
/* nflag (may be NULL): one byte per seed, bit 0 / 1 = the left / right query holds an N (what launch_bin wants to know) */
hipError_t launch_pack(const uint8_t *raw, const bsw_dtask *tasks, const bsw_rawoff *roff, uint32_t bias, uint32_t n, int rev_left,
                       const uint8_t *pac, int64_t l_pac, const bsw_refx *refx, uint64_t *seq, uint8_t *nflag, hipStream_t s);
hipError_t launch_wire_pack(const uint32_t *wire, const bsw_dtask *tasks, const bsw_wireoff *woffs, uint32_t n, uint64_t *seq, hipStream_t s);
/* the group's 16 KiB result batches, written on the device: wout[woffs[t].out_word .. + 4] = R0..R4 of task t, the rest zero */
hipError_t launch_wire_results(const bsw_result *out, const bsw_wireoff *woffs, uint32_t n, uint32_t *wout, size_t wout_words, hipStream_t s);
/* word offsets of tasks[0..n) made absolute (bsw_rebase in bsw_device.h) */
hipError_t launch_rebase(bsw_dtask *tasks, const bsw_rawoff *roff, uint32_t n, const bsw_rebase &rb, hipStream_t s);
int global_class_count();
int global_class_cols(int cls);
hipError_t launch_global(int cls, const bsw_dparams &P, const uint64_t *seq, const bsw_gdtask *tasks, const uint32_t *order, uint32_t n,
                         uint8_t *z, uint32_t *cigars, int max_cigar, bsw_gresult *out, hipStream_t s);
int align_class_count();
int align_class_of(int qlen, int byte_mode);                 /* -1: query too long for the mode */
hipError_t launch_align(int cls, const bsw_dparams &P, const uint64_t *seq, const bsw_adtask *tasks, const uint32_t *order, uint32_t n,
                        unsigned long long *blist, bsw_kswr *out, hipStream_t s);
/* nflag == NULL: the queries' words are read from seq instead (input that arrived packed); keys: n words of scratch */
hipError_t launch_bin(const bsw_binparams &bp, const uint64_t *seq, const uint8_t *nflag, const bsw_dtask *tasks, uint32_t n, uint32_t *bins,
                      uint64_t *keys, uint32_t *order, hipStream_t s);
}  // namespace bsw

#endif
