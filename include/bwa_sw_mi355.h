/*
 * bwa_sw_mi355.h — C ABI of libbwasw_mi355.so
 *
 * MI355X (gfx950) implementation of BWA-MEM's banded affine-gap Smith-Waterman
 * seed-extension path: ksw_extend / ksw_extend2 as driven by mem_chain2aln
 * (left extension, right extension, MAX_BAND_TRY band doubling, clip-vs-extend
 * decision).  Every entry point below computes on the GPU through hand-written
 * HIP kernels; there is NO CPU fallback in this library.  If no gfx950 device
 * (or no usable HIP runtime) is present the calls fail with BSW_E_NODEVICE.
 *
 * What each entry point replaces in the reference (peterpengwei/bwa-mem-sw,
 * FPGA RTL; citations are into that tree):
 *
 *   ksw_extend2 / ksw_extend   the software ABI the accelerator stands in for
 *                              (bwa ksw.h; hardware statement of it:
 *                              sw_pe_array_sw_extend.v:96-123 ports,
 *                              :1639-1705 FSM)
 *   bsw_params                 batch-global words G0/G1
 *                              (sw_pe_array_proc_element.v:816-819, :916-917)
 *   bsw_task                   8-word task header H0..H7 + packed bases
 *                              (sw_pe_array_task_parse.v:1884-1951,
 *                               sw_pe_array_proc_element.v:807-933, :1638-1683)
 *   bsw_result                 5-word result record R0..R4
 *                              (sw_pe_array_proc_element.v:1662-1665, :1190-1199)
 *   bsw_create/.../bsw_wait    CSR + DSM handshake and the TBB/RBB round trip
 *                              (batch_manager.v:208-221, :358-739; tbb.v; rbb.v)
 *   bsw_refbatch_*             the exact 256 KiB task batch / 16 KiB result
 *                              batch wire format (bwa_mem_sw.v:163-170)
 *
 * Base codes: 0..3 = A,C,G,T; 4 = N; one base per byte, exactly as bwa passes
 * them to ksw_extend.  Left-extension query/target must already be reversed by
 * the caller (as mem_chain2aln does before calling ksw_extend2).
 */
#ifndef BWA_SW_MI355_H
#define BWA_SW_MI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes (never abort; the reference signals no errors at all) ---- */
#define BSW_OK            0
#define BSW_E_NODEVICE   (-1)  /* no gfx950 GPU / HIP runtime unusable            */
#define BSW_E_INVAL      (-2)  /* NULL pointer, bad params (e_ins<=0, h0<=0, ...) */
#define BSW_E_LIMIT      (-3)  /* task outside device limits (see BSW_MAX_*)      */
#define BSW_E_HIP        (-4)  /* a HIP runtime call failed                       */
#define BSW_E_NOMEM      (-5)
#define BSW_E_BUSY       (-6)  /* BSW_MAX_INFLIGHT submits in flight already, or a synchronous call while one is */

/* ---- device limits ------------------------------------------------------- */
#define BSW_MAX_QLEN   8191    /* query side length per extension (ksw_extend2 and the batch API; up to 1 023 in registers,
                                  beyond that the eh[] row lives in LDS: bsw_long_kernel.hip)                          */
#define BSW_GLOBAL_MAX_QLEN 1023 /* ksw_global2 / bsw_global_batch                 */
#define BSW_MAX_TLEN   65535   /* target side length per extension                */
#define BSW_MAX_SCORE  (1 << 20) /* h0 + qlen*max(mat) must stay below this       */

/* recurrence variant: H = bwa<=0.7.8 and the RTL
 * (sw_pe_array_sw_extend.v:1797-1798,1863,1866); M = bwa>=0.7.9 ("M? M+q : 0") */
#define BSW_VARIANT_H  0
#define BSW_VARIANT_M  1

typedef struct bsw_params {
    int8_t  mat[25];        /* 5x5 scoring matrix, row = target base (K6)        */
    int8_t  _pad[3];
    int32_t o_del, e_del;   /* G0 [7:0], [15:8]                                  */
    int32_t o_ins, e_ins;   /* G0 [23:16], [31:24]                               */
    int32_t w;              /* G1 [23:16]  opt->w                                */
    int32_t pen_clip5;      /* G1 [7:0]    also end_bonus of the left extension  */
    int32_t pen_clip3;      /* G1 [15:8]   also end_bonus of the right extension */
    int32_t zdrop;          /* not in the RTL (quirk Q3); bwa default 100        */
    int32_t max_band_try;   /* MAX_BAND_TRY, 2 in bwa and in the RTL (:1963)     */
    int32_t variant;        /* BSW_VARIANT_H (default) or BSW_VARIANT_M          */
} bsw_params;

/* One seed = left + right extension (what one RTL processing element handles). */
typedef struct bsw_task {
    const uint8_t *lquery;  /* reversed query[0..qbeg)          (H0 qlen0)       */
    const uint8_t *ltarget; /* reversed reference left of seed  (H0 tlen0)       */
    const uint8_t *rquery;  /* query[qbeg+len..l_query)         (H1 qlen1)       */
    const uint8_t *rtarget; /* reference right of the seed      (H1 tlen1)       */
    int32_t  lqlen, ltlen, rqlen, rtlen;
    int32_t  h0;            /* s->len * a                       (H4)             */
    int32_t  init_score;    /* a->score before the left ext; bwa: -1 (H3 low)    */
    int32_t  qbeg;          /* s->qbeg                          (H3 high)        */
    uint32_t tag;           /* opaque, echoed back              (H7 -> R0)       */
    /* Host-supplied band limits min(max_ins, max_del) per side — the RTL's header words H5 / H6
     * (sw_pe_array_proc_element.v:925,933; applied at sw_pe_array_sw_extend.v:1881,1890).
     * 0 = let the library compute them with bwa's formula from the scoring parameters. */
    int32_t  wlim_l, wlim_r;
} bsw_task;

/* Raw outputs of the last ksw_extend2 pass of one side (K9) + bookkeeping. */
typedef struct bsw_ext {
    int32_t  score, qle, tle, gtle, gscore, max_off;
    int32_t  aw;            /* band width of the last pass, unclamped w<<k (P3)  */
    uint32_t cells;         /* DP cells evaluated over all passes of this side   */
} bsw_ext;

typedef struct bsw_result {
    uint32_t tag;           /* R0                                                */
    int32_t  qb, qe;        /* R1: qb absolute in the query; qe relative to the
                                   seed's query end                              */
    int32_t  rb, re;        /* R2: relative to the seed's reference begin / end
                                   (rb = -tle or -gtle)                          */
    int32_t  score, truesc; /* R3                                                */
    int32_t  w;             /* R4: max(aw_left, aw_right)                        */
    bsw_ext  left, right;   /* not in the RTL record; for parity checking        */
} bsw_result;

/* The pair-level record alone: exactly the information of the RTL's 5-word result record R0..R4
 * (sw_pe_array_proc_element.v:1662-1665,1190-1199) — the first 32 bytes of bsw_result.  A context created with
 * bsw_config.result_format = BSW_RESULT_PAIR hands THESE back from bsw_submit / bsw_submit_packed / bsw_submit_ref
 * (the `out` argument then addresses bsw_pair[n], cast to bsw_result *): a third of the bytes over PCIe and out of
 * the finalize kernel; the per-side bsw_ext records stay in device scratch. */
typedef struct bsw_pair {
    uint32_t tag;
    int32_t  qb, qe, rb, re, score, truesc, w;
} bsw_pair;

/* One plain ksw_extend2 call (no band retry), for batched single extensions. */
typedef struct bsw_ext_task {
    const uint8_t *query, *target;
    int32_t qlen, tlen;
    int32_t w, end_bonus, h0;
} bsw_ext_task;

#define BSW_MAX_DEVICES 16
typedef struct bsw_config {
    int32_t device;         /* HIP device ordinal (used when n_devices == 0)      */
    int32_t kernel;         /* BSW_KERNEL_*                                      */
    int32_t streams;        /* staging slots per device = streams = pipeline threads of bsw_submit (1..8, def 4) */
    int32_t pack_threads;   /* helper threads of the slots (def 8): a chunk's host pass (validate, lay out, count) runs on
                               1 + pack_threads / streams threads side by side, and so does the gather of sequences that
                               are NOT in registered memory into pinned staging */
    size_t  chunk_tasks;    /* tasks per H2D/launch chunk in bsw_submit; 0 (def) = sized per submit by the seeds' work:
                               128 Ki seeds of 131-base sides, more of shorter ones (two chunks in flight fill the GPU) */
    /* One context can drive several GPUs, as the reference's batch manager drives its 4 PE arrays
     * round-robin (batch_manager.v:343-348,418; bwa_mem_sw.v:162): chunk k of a bsw_submit goes to
     * devices[k mod n_devices].  n_devices == 0 means the single `device` above.  The same ordinal
     * may appear more than once (more slots on that GPU). */
    int32_t n_devices;
    int32_t devices[BSW_MAX_DEVICES];
    int32_t timeout_ms;     /* watchdog on every wait for the GPU; 0 (def) = BSW_TIMEOUT_MS from the environment, else 120000
                               (bsw_effective_timeout_ms); on expiry the call fails with BSW_E_HIP and the context is dead
                               (every later call fails fast)  */
    int32_t result_format;  /* BSW_RESULT_FULL (def): bsw_result[n]; BSW_RESULT_PAIR: bsw_pair[n] from the bsw_submit* calls */
    int32_t pin_threads;    /* 0 / 1 (def): the slot and gather threads of a device run on the CPUs of that GPU's NUMA node
                               (/sys/bus/pci/devices/<bdf>/local_cpulist, intersected with the process affinity) and their
                               pinned staging is allocated and first touched from there — one manager next to its arrays
                               (batch_manager.v:745-773); -1: threads and staging are left where the OS puts them */
} bsw_config;               /* 104 bytes in ABI 5 (pin_threads took the tail padding of ABI 4) */

/* ABI of this header.  bsw_config has no size field of its own: bsw_create() reads sizeof(bsw_config) of THIS header, so a
 * caller built against an older, shorter struct must use bsw_create_sized() with ITS sizeof (fields beyond it take their
 * defaults) — or check bsw_abi_version() == BSW_ABI_VERSION at start-up.  History: 3 = 96-byte config; 4 = result_format
 * (104 bytes); 5 = pin_threads in the former padding, bsw_create_sized, bsw_chain_timeouts; 6 = tickets (bsw_submit*_t,
 * bsw_wait_ticket, bsw_test: BSW_MAX_INFLIGHT submits per context where 5 answered BSW_E_BUSY to the second), bsw_host_stats,
 * bsw_upload_raw / bsw_run_staged, bsw_default_config writes timeout_ms = 0 (bsw_effective_timeout_ms). */
#define BSW_ABI_VERSION 6

#define BSW_RESULT_FULL  0
#define BSW_RESULT_PAIR  1

#define BSW_KERNEL_AUTO  0  /* per-bin choice (batch manager)                    */
#define BSW_KERNEL_WAVE  1  /* one wavefront per task, row-synchronous           */
#define BSW_KERNEL_LANE  2  /* one lane per task, inter-task SIMD                */

typedef struct bsw_ctx bsw_ctx;
typedef struct bsw_dev_batch bsw_dev_batch;

/* ---- drop-in scalar ABI (bwa ksw.h).  Thread-safe: concurrent callers (bwa mem -t N worker threads)
 * are coalesced into one device batch per round trip (leader/follower; no per-call allocation).
 * Contract at the edges: h0 <= 0 or qlen <= 0 (outside bwa's assert(h0 > 0) domain) returns
 * max(h0,0) with qle = tle = gtle = 0, gscore = -1, max_off = 0 without touching the GPU; on a device
 * failure the call prints the reason to stderr, writes the same neutral outputs and returns -1. ---- */
int ksw_extend2(int qlen, const uint8_t *query, int tlen, const uint8_t *target,
                int m, const int8_t *mat, int o_del, int e_del, int o_ins, int e_ins,
                int w, int end_bonus, int zdrop, int h0,
                int *qle, int *tle, int *gtle, int *gscore, int *max_off);
int ksw_extend(int qlen, const uint8_t *query, int tlen, const uint8_t *target,
               int m, const int8_t *mat, int gapo, int gape,
               int w, int end_bonus, int zdrop, int h0,
               int *qle, int *tle, int *gtle, int *gscore, int *max_off);
/* recurrence variant used by ksw_extend/ksw_extend2 (process-wide, default H) */
void bsw_set_default_variant(int variant);
/* calls served / device round trips made by the scalar ABI so far (their ratio = mean coalescing factor) */
void bsw_scalar_stats(uint64_t *calls, uint64_t *trips);

/* ---- batch API ------------------------------------------------------------ */
void     bsw_default_params(bsw_params *p);          /* bwa defaults a=1,b=4,o=6,e=1,w=100,clip=5,zdrop=100 */
void     bsw_default_config(bsw_config *c);
int      bsw_device_count(void);                     /* gfx950 devices visible; <=0 if none */
int      bsw_create(const bsw_config *cfg, bsw_ctx **out);
int      bsw_create_sized(const bsw_config *cfg, size_t cfg_size, bsw_ctx **out);   /* cfg_size = the CALLER's sizeof(bsw_config) */
/* Where device k of the context (index into bsw_config.devices[]) sits: PCI address ("0000:c1:00.0"), NUMA node (-1: the
 * kernel does not say) and how many CPUs next to it the context's slot threads are pinned to (0: not pinned). */
int      bsw_device_placement(const bsw_ctx *ctx, int k, char *bdf, size_t bdf_cap, int *numa_node, int *n_cpus);
int      bsw_abi_version(void);                      /* BSW_ABI_VERSION the library was built with */
/* The watchdog (ms) a context created from cfg would run with: cfg->timeout_ms when > 0, else the environment variable
 * BSW_TIMEOUT_MS when it holds a positive number, else 120000.  bsw_default_config writes 0.  Host only. */
int      bsw_effective_timeout_ms(const bsw_config *cfg);
/* How many of the launch chain's waiting waves gave up at their 20 ms deadline so far (the chain's flag is a
 * scheduling hint, a follower released early is still correct).  Non-zero where kernels are run one at a time
 * (rocprofv3 --pmc, HIP_LAUNCH_BLOCKING, AMD_SERIALIZE_KERNEL); 0 in normal operation.  Synchronises the context. */
int      bsw_chain_timeouts(bsw_ctx *ctx, uint64_t *n);
void     bsw_destroy(bsw_ctx *ctx);
const char *bsw_last_error(const bsw_ctx *ctx);      /* text of the last failure  */

/* ---- host memory the GPU can DMA directly.  Sequences and result arrays that live in memory obtained
 * from bsw_host_alloc (or registered with bsw_host_register) cross PCIe with NO host copy: bsw_submit
 * DMAs the byte-per-base arena of a chunk as it is and packs / bins it on the GPU.  Anything else is first
 * gathered into pinned staging by `pack_threads` host threads (correct, but host-bound).  Process-wide. ---- */
void    *bsw_host_alloc(size_t bytes);
void     bsw_host_free(void *p);
int      bsw_host_register(void *p, size_t bytes);
int      bsw_host_unregister(void *p);

/* Asynchronous: validates lengths, DMAs the raw sequences to the device in chunks (chunk k ->
 * device k mod n_devices), packs + bins them there, launches, copies results back into out[] in
 * TASK ORDER.  out[], tasks[] and the task sequences must stay valid until the submit has been waited for.
 * Up to BSW_MAX_INFLIGHT submits may be in flight per context (the reference's manager keeps four task batches
 * going: batch_manager.v:343-348 request bits, :434-435 busy bitmap); their chunks run through the context's slots in
 * submit order, so the tail of one overlaps the head of the next.  One more than that answers BSW_E_BUSY.
 * bsw_wait waits for ALL of them and returns the first failure in submit order. */
#define BSW_MAX_INFLIGHT 4
typedef uint64_t bsw_ticket;            /* names one submit; never 0 */
int      bsw_submit(bsw_ctx *ctx, const bsw_params *p, const bsw_task *tasks, size_t n, bsw_result *out);
int      bsw_wait(bsw_ctx *ctx);
/* the same submit, handing back its ticket (ticket may be NULL) */
int      bsw_submit_t(bsw_ctx *ctx, const bsw_params *p, const bsw_task *tasks, size_t n, bsw_result *out, bsw_ticket *ticket);
/* wait for ONE submit and collect it: its error code, BSW_E_INVAL for a ticket that is not in flight */
int      bsw_wait_ticket(bsw_ctx *ctx, bsw_ticket ticket);
/* the host's status poll (the busy nibble at DSM + 0x40, batch_manager.v:844-854): 1 = complete (out[] is filled; collect the
 * submit and its error code with bsw_wait_ticket / bsw_wait), 0 = in flight, < 0 = no such ticket.  Never blocks. */
int      bsw_test(bsw_ctx *ctx, bsw_ticket ticket);
int      bsw_inflight(bsw_ctx *ctx);    /* submits not collected by a wait yet */
/* What the host side of the streaming path has cost so far: CPU time of the context's slot threads (validate, lay out,
 * count, start DMAs, poll) and of the gather helpers (sequences outside registered memory), and the volume moved.  out_size
 * = the caller's sizeof(bsw_stats). */
typedef struct bsw_stats {
    uint64_t slot_cpu_ns, helper_cpu_ns;    /* CLOCK_THREAD_CPUTIME_ID, summed over threads */
    uint64_t seeds, chunks, submits;
    uint64_t h2d_bytes, d2h_bytes;
    uint64_t slot_threads;
} bsw_stats;
int      bsw_host_stats(bsw_ctx *ctx, bsw_stats *out, size_t out_size);
/* The same for callers that keep sequences 4-BIT PACKED: the bsw_task pointers then address uint64 words, 16 bases
 * each (base k in bits [4k, 4k+3]; codes 0-3 = ACGT, 4-7 = N), every sequence starting on an 8-byte boundary; the
 * lengths stay in bases; the unused nibbles behind a sequence's last base may hold anything.  This is the device's own layout and the encoding the reference ships over its link (8 bases
 * per 32-bit word, first base in the top nibble there: sw_pe_array_proc_element.v:1638,1677-1683).  Words in registered
 * memory are DMA'd straight into the sequence buffer — no pack kernel, ~0.6x the PCIe bytes per seed of bsw_submit.
 * bsw_pack_bases() packs one byte-per-base sequence; bsw_pack_tasks() a whole task array. */
int      bsw_submit_packed(bsw_ctx *ctx, const bsw_params *p, const bsw_task *tasks, size_t n, bsw_result *out);
int      bsw_submit_packed_t(bsw_ctx *ctx, const bsw_params *p, const bsw_task *tasks, size_t n, bsw_result *out, bsw_ticket *ticket);
/* tasks[0..n) (byte per base) -> out[0..n) with the sequences packed into `arena` (8-byte aligned, `cap` bytes);
 * returns the bytes used or <0 (BSW_E_NOMEM: arena too small).  bsw_pack_tasks_bound() = a sufficient `cap`. */
int64_t  bsw_pack_tasks(const bsw_task *tasks, size_t n, uint64_t *arena, size_t cap, bsw_task *out);
size_t   bsw_pack_tasks_bound(const bsw_task *tasks, size_t n);
/* Batched plain ksw_extend2 (one pass each, w/end_bonus/h0 per task); synchronous. */
int      bsw_extend_batch(bsw_ctx *ctx, const bsw_params *p, const bsw_ext_task *tasks, size_t n, bsw_ext *out);

/* ---- banded global alignment with CIGAR: bwa's ksw_global2 / ksw_global (ksw.c), the
 * Smith-Waterman user after seed extension inside mem_reg2aln (bwa_gen_cigar2).  Not in the reference RTL — it
 * belongs to the host software named at reference README.md:7-18.  CIGAR encoding is BAM's: len << 4 | op,
 * op 0 = M, 1 = I, 2 = D. ---- */
typedef struct bsw_gtask {
    const uint8_t *query, *target;   /* codes 0..4, one per byte */
    int32_t qlen, tlen, w;           /* w = band half-width */
    int32_t _pad;
} bsw_gtask;
typedef struct bsw_gresult {
    int32_t score;                   /* eh[qlen].h of the last row, exactly as bwa returns it */
    int32_t n_cigar;                 /* CIGAR operations written; < 0: -n operations did not fit max_cigar */
} bsw_gresult;
/* Batched ksw_global2 on the GPU.  cigars (may be NULL: scores only) receives max_cigar words per task. */
int      bsw_global_batch(bsw_ctx *ctx, const bsw_params *p, const bsw_gtask *tasks, size_t n, int max_cigar,
                          bsw_gresult *res, uint32_t *cigars);
/* drop-in scalar ABI (bwa ksw.h): *cigar is malloc'ed, the caller frees it; n_cigar / cigar may be NULL */
int ksw_global2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat,
                int o_del, int e_del, int o_ins, int e_ins, int w, int *n_cigar, uint32_t **cigar);
int ksw_global(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat,
               int gapo, int gape, int w, int *n_cigar, uint32_t **cigar);

/* ---- local alignment with start / second-best search (second half: bwa ksw.h ksw_align2, the
 * Smith-Waterman of mate rescue, mem_matesw; like ksw_global2 it lives in the reference's host software, the
 * bwa-0.7.8 tree named at /root/reference/README.md:7-18, not in the RTL) ---------------------------------
 * Results are those of bwa's striped SSE2 code (ksw_u8 when KSW_XBYTE is set, else ksw_i16), bit for bit: score
 * (255 = the 8-bit run saturated, call again without KSW_XBYTE), te/qe (end on target / query, inclusive),
 * score2/te2 (best end at least ceil(score/max) rows away, needs KSW_XSUBO | threshold), tb/qb (start, needs
 * KSW_XSTART; -1 when the start pass did not run or disagrees).  xtra = flags | 16-bit threshold, as in bwa. */
#define KSW_XBYTE  0x10000
#define KSW_XSTOP  0x20000
#define KSW_XSUBO  0x40000
#define KSW_XSTART 0x80000
#define BSW_ALIGN_MAX_QLEN 1024
typedef struct bsw_kswr {            /* = bwa's kswr_t */
    int32_t score, te, qe, score2, te2, tb, qb;
} bsw_kswr;
typedef struct bsw_atask {
    const uint8_t *query, *target;   /* codes 0..4, one per byte */
    int32_t qlen, tlen;              /* qlen <= BSW_ALIGN_MAX_QLEN */
    int32_t xtra;
    int32_t _pad;
} bsw_atask;
/* Batched ksw_align2 on the GPU (m = 5; p supplies mat and the four gap penalties). */
int      bsw_align_batch(bsw_ctx *ctx, const bsw_params *p, const bsw_atask *tasks, size_t n, bsw_kswr *out);
/* drop-in scalar ABI (bwa ksw.h).  qry (bwa's query-profile cache) is not used: pass NULL, or a pointer whose target
 * stays NULL. */
#ifndef __AC_KSW_H
typedef struct { int score; int te, qe; int score2, te2; int tb, qb; } kswr_t;
#endif
kswr_t ksw_align2(int qlen, uint8_t *query, int tlen, uint8_t *target, int m, const int8_t *mat,
                  int o_del, int e_del, int o_ins, int e_ins, int xtra, void **qry);
kswr_t ksw_align(int qlen, uint8_t *query, int tlen, uint8_t *target, int m, const int8_t *mat,
                 int gapo, int gape, int xtra, void **qry);

/* ---- device-resident batches (inputs in HBM before the timed region) ------- */
int      bsw_upload(bsw_ctx *ctx, const bsw_params *p, const bsw_task *tasks, size_t n, bsw_dev_batch **out);
int      bsw_upload_packed(bsw_ctx *ctx, const bsw_params *p, const bsw_task *tasks, size_t n, bsw_dev_batch **out);   /* 4-bit packed sequences, as bsw_submit_packed */
int      bsw_run(bsw_ctx *ctx, bsw_dev_batch *b);          /* enqueue kernels only        */
/* bsw_upload that also KEEPS the byte-per-base sequences (as they crossed PCIe) in HBM, and bsw_run_staged = the whole device
 * side of one chunk of bsw_submit on such a batch: pack (bytes -> 16 bases per uint64) + bin (counting sort into the launch
 * lists) + the DP kernels — the batch manager's work from "task batch landed" to "result batch ready" (tbb.v:110-123,
 * sw_pe_array_task_parse.v:1600-1648, rbb.v:219-224), inputs resident in HBM.  bsw_run on such a batch still runs the DP
 * kernels alone.  bsw_run_history2 splits every run since the last call into total / pack + bin milliseconds (HIP events on
 * the library's stream). */
int      bsw_upload_raw(bsw_ctx *ctx, const bsw_params *p, const bsw_task *tasks, size_t n, bsw_dev_batch **out);
int      bsw_run_staged(bsw_ctx *ctx, bsw_dev_batch *b);
int      bsw_run_history2(bsw_ctx *ctx, float *total_ms, float *staging_ms /* may be NULL */, int cap);
int      bsw_sync(bsw_ctx *ctx);                            /* hipStreamSynchronize        */
int      bsw_download(bsw_ctx *ctx, bsw_dev_batch *b, bsw_result *out); /* task order      */
int      bsw_batch_info(const bsw_dev_batch *b, uint64_t *n_tasks, uint64_t *in_bytes, uint64_t *out_bytes, uint64_t *n_launches);
/* the launch order the device-side binning produced for a resident batch (same layout as bsw_plan_batch:
 * order[] capacity 4*n+16, seg[BSW_PLAN_SEGS+1]); for tests and tools */
int      bsw_batch_order(bsw_ctx *ctx, const bsw_dev_batch *b, uint32_t *order, uint32_t *seg);
/* time of the kernels of the last bsw_run, measured with hipEvents on the
 * library's own stream; valid after bsw_sync.                                  */
int      bsw_last_run_ms(bsw_ctx *ctx, float *ms);
/* kernel time (ms) of every bsw_run since the previous call, oldest first; returns the
 * count written (<= cap) and resets the history.  Synchronises on the recorded events.   */
int      bsw_run_history(bsw_ctx *ctx, float *ms, int cap);
void     bsw_free_batch(bsw_ctx *ctx, bsw_dev_batch *b);

/* ---- batch plan (host only, no GPU needed): how the batch manager cuts tasks[0..n) into launches.
 * The device sorts the seeds (bsw_stage_kernel.hip); the host only counts them per class, with the same
 * class functions, to size the launches.  order[] (capacity 4*n+16, may be NULL) receives a launch order
 * built on the host with the device's rules (lane sides: queries with an N first, each part longest first, left
 * sides of one length by h0 bucket — 8 buckets over the chunk's h0 range; inside one bin the device's order is arbitrary);
 * seg[] receives BSW_PLAN_SEGS+1 offsets into order[]:
 * segments 0..7 = wave-per-task classes (64,128,192,256,512,1024,2048,8192 columns), 8 = all lane seeds,
 * 9..16 = lane left sides per lane class, 17..24 = lane right sides per lane class, 25 = redo list space.
 * kernel = BSW_KERNEL_*.  Returns the number of sequence words the batch needs, or <0. ---- */
#define BSW_PLAN_SEGS 26
int64_t  bsw_plan_batch(const bsw_params *p, const bsw_task *tasks, size_t n, int kernel, int pack_threads,
                        uint32_t *order, uint32_t *seg /*[BSW_PLAN_SEGS+1]*/);

/* ---- reference wire format (bwa_mem_sw.v:163-170) ----------- */
#define BSW_REFBATCH_IN_WORDS   65536   /* 256 KiB task batch (tbb.v:59)         */
#define BSW_REFBATCH_OUT_WORDS  4096    /* 16 KiB result batch (rbb.v:59)        */
#define BSW_REFBATCH_MAX_TASKS  819     /* floor(4096/5) five-word records       */
/* Encode up to n tasks; returns the number of tasks that fit (>=0) or <0.
 * Tasks whose fields exceed the RTL's port widths (quirk Q1) stop the batch.   */
int      bsw_refbatch_encode(const bsw_params *p, const bsw_task *tasks, size_t n, uint32_t *words /*[65536]*/);
/* Decode a task batch; sequences are unpacked into seqbuf (byte per base).     */
int      bsw_refbatch_decode(const uint32_t *words, bsw_params *p, bsw_task *tasks, size_t max_tasks,
                             uint8_t *seqbuf, size_t seqbuf_len);
int      bsw_refbatch_encode_results(const bsw_result *res, size_t n, uint32_t *words /*[4096]*/);
int      bsw_refbatch_decode_results(const uint32_t *words, size_t n, bsw_result *res);
/* Run one 256 KiB task batch end to end on the GPU and fill the 16 KiB result
 * batch — what one RTL PE array does between task_start and TestCmp.           */
int      bsw_refbatch_run(bsw_ctx *ctx, const uint32_t *in_words, uint32_t *out_words, int variant, int zdrop);
/* Queue a task batch (asynchronous; the reference keeps 4 in flight, batch_manager.v:418,745-773; here up to
 * BSW_REFBATCH_MAX_INFLIGHT).  in_words / out_words must stay valid until bsw_refbatch_wait, which runs
 * everything queued as ONE device batch (the nibble streams are unpacked on the GPU), fills every out_words
 * in TASK ORDER (the RTL emits completion order and relies on the tag, sw_pe_array_fill_resulBuf.v:377-429)
 * and returns the number of batches completed or <0.  All batches of one wait share variant / zdrop.
 * The header's H5/H6 (max_ins/max_del) are honoured as the band limits, as in the RTL. */
#define BSW_REFBATCH_MAX_INFLIGHT 256
int      bsw_refbatch_submit(bsw_ctx *ctx, const uint32_t *in_words, uint32_t *out_words);
int      bsw_refbatch_wait(bsw_ctx *ctx, int variant, int zdrop);

/* ---- mem_chain2aln caller glue: what bwamem.c does either side of ksw_extend2.
 * Reference coordinates are bwa's: [0, l_pac) forward strand, [l_pac, 2*l_pac) reverse complement. ---- */
typedef struct bsw_seed {           /* mem_seed_t */
    int64_t rbeg;
    int32_t qbeg, len;
} bsw_seed;
typedef struct bsw_alnreg {         /* the fields of mem_alnreg_t this path produces */
    int64_t rb, re;
    int32_t qb, qe;
    int32_t score, truesc, w;
} bsw_alnreg;
/* cal_max_gap(opt, qlen): longest gap a qlen-base flank can pay for, capped at 2w */
int      bsw_cal_max_gap(const bsw_params *p, int qlen);
/* the chain's reference window [rmax[0], rmax[1]) as mem_chain2aln computes it from all seeds of a chain */
int      bsw_chain_window(const bsw_params *p, const bsw_seed *seeds, int n_seeds, int l_query, int64_t l_pac, int64_t rmax[2]);
/* bytes of scratch bsw_seed_to_task needs for the reversed left query + left target */
size_t   bsw_seed_scratch_bytes(const bsw_seed *s, int64_t rmax0);
/* one seed -> one task.  rseq = reference bases of [rmax0, rmax1) (bns_get_seq), query = the read (codes 0..4).
 * The left query/target are written reversed into scratch; right ones point into query / rseq. */
int      bsw_seed_to_task(const bsw_params *p, const bsw_seed *s, int l_query, const uint8_t *query,
                          int64_t rmax0, int64_t rmax1, const uint8_t *rseq,
                          uint8_t *scratch, size_t scratch_len, uint32_t tag, bsw_task *t);
/* result record -> alignment region in read / reference coordinates */
int      bsw_result_to_alnreg(const bsw_seed *s, const bsw_result *r, bsw_alnreg *a);
/* bns_get_seq: bases of [beg, end) from a 2-bit packed reference (4 bases per byte, first base in the top bits);
 * returns the number of bases written, 0 if the range bridges the forward/reverse boundary */
int64_t  bsw_pac_get_seq(int64_t l_pac, const uint8_t *pac, int64_t beg, int64_t end, uint8_t *dst);

/* ---- seeds against a DEVICE-RESIDENT reference: the 2-bit pac is uploaded once, the
 * extension targets are fetched on the GPU (bns_get_seq semantics, both strands, left side reversed),
 * so only the reads travel over PCIe. ---- */
typedef struct bsw_ref bsw_ref;
typedef struct bsw_ref_task {
    const uint8_t *query;   /* the whole read, codes 0..4                                  */
    int32_t  l_query;
    int32_t  init_score;    /* -1 in bwa                                                    */
    bsw_seed seed;          /* rbeg in [0, 2*l_pac), qbeg, len                              */
    int64_t  rmax0, rmax1;  /* reference window of the chain (bsw_chain_window)             */
    uint32_t tag;
    uint32_t _pad;
} bsw_ref_task;
int      bsw_ref_upload(bsw_ctx *ctx, const uint8_t *pac, int64_t l_pac, bsw_ref **out);
void     bsw_ref_free(bsw_ctx *ctx, bsw_ref *ref);
/* device-resident batch whose targets come from `ref`; then bsw_run / bsw_download as usual */
int      bsw_upload_ref(bsw_ctx *ctx, const bsw_params *p, const bsw_ref *ref, const bsw_ref_task *tasks, size_t n, bsw_dev_batch **out);
/* streaming form (like bsw_submit; finish with bsw_wait): chunk k -> device k mod n_devices, every device holding
 * its own copy of the reference (bsw_ref_upload puts one on each GPU of the context).  Only the reads cross PCIe
 * (~1/2.5 of the bytes of bsw_submit at 150 bp); reads in bsw_host_alloc / registered memory are DMA'd as they are. */
int      bsw_submit_ref(bsw_ctx *ctx, const bsw_params *p, const bsw_ref *ref, const bsw_ref_task *tasks, size_t n, bsw_result *out);
int      bsw_submit_ref_t(bsw_ctx *ctx, const bsw_params *p, const bsw_ref *ref, const bsw_ref_task *tasks, size_t n, bsw_result *out, bsw_ticket *ticket);
/* convenience: upload_ref + run + download + free (synchronous) */
int      bsw_extend_ref(bsw_ctx *ctx, const bsw_params *p, const bsw_ref *ref, const bsw_ref_task *tasks, size_t n, bsw_result *out);

/* ---- device sequence format: 4 bits per base, 16 bases per uint64, base k of a word in bits [4k,4k+3];
 * codes > 4 are stored as 4 (N).  words must hold (len+15)/16 entries.  Returns 1 if an N was seen. ---- */
int      bsw_pack_bases(const uint8_t *bases, int len, uint64_t *words);

/* ---- synthetic workload generator (no genome in the image) --- */
typedef struct bsw_synth_spec {
    uint64_t seed;
    int32_t  read_len;      /* 150 or 250                                        */
    int32_t  seed_len_min, seed_len_max; /* seed length ~ U[min,max]             */
    int32_t  seed_at_start; /* 1: seed = read[0:seed_len) (right extension only) */
    double   sub_rate, indel_rate;       /* per-base                             */
    double   n_rate;        /* fraction of bases turned into N                    */
    double   junk_frac;     /* fraction of tasks whose flanks are unrelated       */
    int32_t  a;             /* match score (for h0 = seed_len*a)                  */
    int32_t  w;             /* band, caps tlen = qlen + min(max_gap, 2w)          */
    int32_t  o, e;          /* gap penalties for cal_max_gap                      */
} bsw_synth_spec;
/* Fills tasks[0..n) and the arena; returns bytes of arena used or <0.          */
int64_t  bsw_synth_generate(const bsw_synth_spec *s, size_t n, bsw_task *tasks, uint8_t *arena, size_t arena_len);
size_t   bsw_synth_arena_bound(const bsw_synth_spec *s, size_t n);
/* Synthetic GENOME + reads for the device-resident-reference path: fills pac ((l_pac+3)/4 bytes, bwa's .pac layout)
 * with i.i.d. bases and tasks[0..n) with forward-strand reads of read_len bases (arena: n*read_len bytes), each with
 * one exact seed and flanks derived from the genome at the spec's error rates; rmax = bsw_chain_window of the seed. */
int64_t  bsw_synth_ref_generate(const bsw_synth_spec *s, const bsw_params *p, int64_t l_pac, uint8_t *pac, size_t n,
                                bsw_ref_task *tasks, uint8_t *arena, size_t arena_len);

#ifdef __cplusplus
}
#endif
#endif /* BWA_SW_MI355_H */
