/*
 * bsw_device.h — device-side data layout shared by the host batch manager and the
 * HIP kernels (internal; the public C ABI is include/bwa_sw_mi355.h).
 *
 * HBM layout of one device batch
 *   seq   : uint64[]   all sequences, 4 bits per base, 16 bases per uint64, base k of
 *                      a word in bits [4k,4k+3]; every sequence starts on a word
 *                      boundary; codes 0..3 = ACGT, 4 = N (anything >4 is stored as 4)
 *   tasks : bsw_dtask[] one record per seed (the RTL's 8-word header H0..H7,
 *                      sw_pe_array_proc_element.v:807-933, widened past its 8-bit limits)
 *   order : uint32[]   launch order -> task index (bins sorted by the batch manager)
 *   out   : bsw_result[] indexed by task index (task order, not completion order —
 *                      the RTL's fill_resulBuf emits completion order and needs the tag)
 */
#ifndef BSW_DEVICE_H
#define BSW_DEVICE_H

#include <stdint.h>
#include "../../include/bwa_sw_mi355.h"

typedef struct bsw_dtask {
    uint32_t lq_off, lt_off, rq_off, rt_off;   /* word offsets into seq              */
    uint16_t lqlen, rqlen, ltlen, rtlen;
    uint16_t wlim_l, wlim_r;                   /* min(max_ins,max_del) per side (H5/H6) */
    int32_t  h0, init_score, qbeg;
    uint32_t tag;
} bsw_dtask;                                   /* 44 bytes */

typedef struct bsw_dparams {
    int8_t  mat[25];
    int8_t  pad[3];
    int32_t o_del, e_del, o_ins, e_ins;
    int32_t w, pen_clip5, pen_clip3, zdrop, max_band_try;
} bsw_dparams;

/* one target to fetch from the device-resident 2-bit reference (bsw_fetch_kernel.hip) */
typedef struct bsw_fetch_desc {
    int64_t  x0;          /* coordinate of base 0 in bwa's [0, 2*l_pac) space */
    uint32_t dst_word;    /* word offset into seq */
    uint32_t tlen;
    int32_t  dir;         /* +1 right-extension target, -1 left-extension target (reversed) */
    int32_t  pad;
} bsw_fetch_desc;

#define BSW_KEY_BITS 10                        /* column index bits in the arg-max key */

#endif
