/*
 * ksw_extend_ref.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * Scalar C restatement of the algorithm the reference RTL implements
 * (peterpengwei/bwa-mem-sw: sw_pe_array_sw_extend.v = ksw_extend2 + band
 * retry; sw_pe_array_proc_element.v = mem_chain2aln left/right driver),
 * resolved to the canonical CPU semantics of SURVEY.md §8a (RTL quirks Q1-Q7
 * are NOT reproduced).
 *
 * PARITY UNPINNED: the reference ships no golden vectors, no tests, and its
 * software counterpart (bwa-0.7.8 ksw.c / bwamem.c, repository
 * peterpengwei/bwa-mem-quickassist) is not in this image and cannot be
 * fetched.  The restatement is pinned instead by analytic known-answer tests,
 * an independent full-matrix numpy DP (oracle/py/full_dp.py) and property
 * tests — see tests/ and DESIGN.md.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * call into this directory.
 */
#ifndef KSW_EXTEND_REF_H
#define KSW_EXTEND_REF_H

#include <stdint.h>
#include <stddef.h>
#include "../include/bwa_sw_mi355.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ksw_extend2 with selectable recurrence variant and an exact cell counter
 * (cells += end-beg for every row iterated).  cells may be NULL. */
int ksw_extend2_ref(int qlen, const uint8_t *query, int tlen, const uint8_t *target,
                    int m, const int8_t *mat, int o_del, int e_del, int o_ins, int e_ins,
                    int w, int end_bonus, int zdrop, int h0,
                    int *qle, int *tle, int *gtle, int *gscore, int *max_off,
                    int variant, uint64_t *cells);

int ksw_extend_ref(int qlen, const uint8_t *query, int tlen, const uint8_t *target,
                   int m, const int8_t *mat, int gapo, int gape,
                   int w, int end_bonus, int zdrop, int h0,
                   int *qle, int *tle, int *gtle, int *gscore, int *max_off,
                   int variant, uint64_t *cells);

int ksw_extend2_wlim_ref(int qlen, const uint8_t *query, int tlen, const uint8_t *target,
                         int m, const int8_t *mat, int o_del, int e_del, int o_ins, int e_ins,
                         int w, int end_bonus, int zdrop, int h0,
                         int *qle, int *tle, int *gtle, int *gscore, int *max_off,
                         int variant, uint64_t *cells, int wlim);

/* Strong CPU baseline (oracle/ksw_extend_avx2.c): the same batch, 16 seeds per AVX2 register; same bytes out. */
void bsw_pair_batch_avx2(const bsw_params *p, const bsw_task *tasks, size_t n, bsw_result *out, int nthreads);

/* bwa's banded global alignment with CIGAR (oracle/ksw_global_ref.c).  *cigar_ is malloc'ed. */
int ksw_global2_ref(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat,
                    int o_del, int e_del, int o_ins, int e_ins, int w, int *n_cigar_, uint32_t **cigar_, uint64_t *cells_);

/* mem_chain2aln left/right extension + MAX_BAND_TRY + clip decision for one seed. */
void bsw_pair_ref(const bsw_params *p, const bsw_task *t, bsw_result *r);

/* Batch drivers (pthreads over tasks); nthreads<=1 runs inline. */
void bsw_pair_batch_ref(const bsw_params *p, const bsw_task *tasks, size_t n, bsw_result *out, int nthreads);
void bsw_ext_batch_ref(const bsw_params *p, const bsw_ext_task *tasks, size_t n, bsw_ext *out, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
