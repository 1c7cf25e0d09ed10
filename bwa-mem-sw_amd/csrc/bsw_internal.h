/*
 * bsw_internal.h — what the host-side translation units of libbwasw_mi355.so share (internal; the public C ABI is
 * include/bwa_sw_mi355.h): the context and staging types, the error channel, and the handful of functions that cross
 * file boundaries.
 *   bsw_ctx.hip     context, registered host memory, parameter validation, the device sequence format on the host
 *   bsw_batch.hip   the batch manager: host pass, device staging, kernel launches, resident batches, streaming pipeline
 *   bsw_scalar.hip  batched plain ksw_extend2 and the drop-in scalar entry points' shared queue
 *   bsw_wire.hip    the reference's 256 KiB / 16 KiB wire format end to end (F1)
 *   bsw_f4.hip      ksw_global2 / ksw_align2 hosts (F4)
 * Everything here has hidden visibility: the shared object exports the C ABI only.
 */
#ifndef BSW_INTERNAL_H
#define BSW_INTERNAL_H

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <pthread.h>
#include <sched.h>

#include "bsw_device.h"
#include "bsw_stage.h"

#define BSW_LOCAL __attribute__((visibility("hidden")))

struct bsw_ref {
    std::vector<uint8_t *> d_pac;     /* one copy per device of the context that uploaded it (index = position in devs) */
    int64_t l_pac = 0;
};

/* BSW_KERNEL_AUTO: a lane launch costs one wave's full duration (1.0 ms for a 131-column side) however few seeds it
 * holds, and a chunk with both sides pays it twice; the general kernels scale with the seed count.  Measured crossovers,
 * device-resident (profiles/r4/crossover_general_kernels.json): one-sided 131 x 257 seeds 27 k (17.4 k before the
 * four-seeds-per-wavefront kernel took the long classes), PE mixed bins ~50 k (two lane launches of 1.05 ms against 36 ns
 * per seed).  So: lane bins from LANE_AUTO_MIN eligible seeds PER LAUNCHED SIDE. */
#define LANE_AUTO_MIN 26000
inline bool lane_bins_pay(uint32_t n_lane, const uint32_t *cl, const uint32_t *cr)
{
    uint32_t l = 0, r = 0;
    for (int c = 0; c < BSW_MAX_LANE_CLASSES; ++c) { l += cl[c]; r += cr[c]; }
    const uint32_t sides = (l ? 1u : 0u) + (r ? 1u : 0u);
    return n_lane >= (uint32_t)LANE_AUTO_MIN * (sides ? sides : 1u);
}
/* Between the two sits bsw_lane2g_kernel (a seed pair per group of eight lanes, 16 seeds per wavefront): its launch lasts
 * ~0.55 ms per side where a lane launch lasts 1.0 and the general kernels need a wavefront per seed.  Measured, device-resident
 * (profiles/r6/crossover_group.json; ms per batch, general / group / lane):
 *     PE mixed bins   32 k seeds 1.28 / 1.09 / 1.97    49 k 1.83 / 1.28 / 1.97    65 k 2.39 / 1.44 / 1.96    131 k 4.59 / 2.13 / 1.95
 *     150 bp one bin  13 k seeds 0.66 / 0.57 / 0.98    24 k 0.93 / 0.89 / 0.98    32 k 1.36 / 0.90 / 0.99     49 k 1.77 / 1.25 / 0.99
 * Both workloads cross at the same WORK per launch — the sum of the query lengths of the sides it holds: the group kernel from
 * ~1.5 M bases (28 k PE sides of ~55 bases, 11 k sides of 131), the lane kernels from ~5 M (100 k PE seeds, 38 k one-bin seeds).
 * A chunk of two-sided seeds that does not fill the machine runs BOTH sides of every seed in one launch (fuse_lists: one wave
 * lifetime instead of two; the launch then holds both sides' bases): PE mixed bins general / group fused / lane fused / AUTO
 *     16 k seeds 0.75 / 0.69 / 1.23 / 0.69    49 k 1.85 / 1.16 / 1.22 / 1.17    131 k 4.62 / 2.62 / 1.33 / 1.33    262 k 9.08 / 4.69 / 1.65 / 1.64 */
#define GROUP_WORK_MIN 1500000ull
#define LANE_WORK_MIN  5000000ull
#define GROUP_WORK_MIN_WIDE 2200000ull   /* ... a chunk with sides beyond 136 columns (four stripes, two waves per SIMD: 250 bp reads cross at ~10.5 k seeds, 1.30 ms) */
#define GROUP_FUSE_MAX 49152u        /* 8-bit lane seeds up to which a group chunk runs both sides in one launch (fuse_lists): 3 waves per SIMD */
#define GROUP_FUSE_MAX_WIDE 98304u   /* ... when the chunk has sides beyond 136 columns (250 bp reads): the lane kernels then need four chained launches,
                                        5.7 - 6.0 ms, and the fused group launch stays ahead up to ~110 k seeds (65 k: 3.5 ms, 131 k: 6.4) */
inline uint32_t group_fuse_max(bool wide) { return wide ? GROUP_FUSE_MAX_WIDE : GROUP_FUSE_MAX; }
#define LANE_FUSE_MAX  262144u       /* ... and a lane-kernel chunk (bsw_lane2_kernel<17, 2, ., ., true>): 2 waves per SIMD */
#define NSPLIT_MAX     262144u       /* lane seeds up to which a chunk's queries with an N go to the general kernel (bsw_binparams.nsplit) ... */
#define NLIST_WORK_MAX 1800000ull    /* ... as long as a sample of the chunk says the list stays below this many QUERY BASES (the general kernel
                                        takes ~0.4 ms per million beside a lane launch of 1.0 - 1.7 ms): measured on / off with bench.py's N rate,
                                        ten times a sequencer's — 131 k PE seeds (1.6 M bases on the list) 1.57 / 2.29 ms, 262 k (3.2 M) 2.86 / 2.48,
                                        131 k one-bin seeds (2.1 M) 1.77 / 1.55, 65 k (1.0 M) 1.04 / 1.54; a sequencer's rate: 262 k PE seeds 1.73 / 2.63 */
#define NSPLIT_MAX_BLIND 131072u     /* ... and without a sample (the wire format's nibble streams) */
#define RAW_SLACK 64                 /* bytes the pack kernel may read past the last sequence */
#define RAW_FRONT 32                 /* ... and in front of the first one (reversed left queries) */

/* How one batch is cut into kernel launches (all offsets index the device `order` array).
 *   [wave classes][lane seeds, any order][lane left sides by qlen][lane right sides by qlen][redo list] + counter */
struct batch_plan {
    uint32_t wave_start[BSW_MAX_WAVE_CLASSES + 1] = {0};
    uint32_t lane_all_off = 0, lane_all_cnt = 0;
    uint32_t laneL_off[BSW_MAX_LANE_CLASSES + 1] = {0}, laneR_off[BSW_MAX_LANE_CLASSES + 1] = {0};
    uint32_t redo_off = 0;
    uint32_t order_len = 0;          /* entries before the redo counter */
    int redo_cls = 0;
    int lane_group = 0;              /* 1: the lane launches of this chunk run the eight-lanes-per-seed-pair kernel (bsw_fin.group);
                                        2: ONE such launch for both sides of every seed, over all 8-bit left lists (fuse_lists) */
    int fused_cls = -1;              /* >= 0: the chunk's 8-bit lane seeds run both sides in ONE launch (the group kernel when lane_group == 2,
                                        else bsw_lane2_kernel's fused instantiation) of this class — it holds every side of the chunk —
                                        over order[fused_off .. fused_off + fused_cnt): all 8-bit left lists */
    uint32_t fused_off = 0, fused_cnt = 0;
    int nsplit = 0;                  /* 1: the chunk's 8-bit lane seeds with an N in a query sit on order[nlist_off ..] for the general kernel
                                        (bsw_binparams.nsplit); their number: the device word order[nlist_cnt_at] */
    uint32_t nlist_off = 0, nlist_cnt_at = 0;
    /* dep[lc] bit rc: some seed has its left side in lane class lc and its right side in lane class rc — the right-side
     * launch of class rc then has to wait for the left-side launch of class lc (h0 of the right extension is the score
     * after the left one, sw_pe_array_proc_element.v:1671).  All ones = not known. */
    uint8_t dep[BSW_MAX_LANE_CLASSES] = {0xff, 0xff, 0xff, 0xff};
};
static_assert(BSW_MAX_LANE_CLASSES == 4, "batch_plan::dep initialiser");

/* error text travels with the thread that produced it; the context keeps the last one */
struct errs {
    std::string msg;
};

inline int fail(errs &e, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    e.msg = buf;
    return code;
}

#define HIPCHK(e, call)                                                                           \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(e, BSW_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

/* ---- growable buffers ------------------------------------------------------- */
template <class T>
struct dbuf {                         /* device */
    T *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t need)
    {
        if (cap >= need) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = need + need / 4 + 16;
        hipError_t e = hipMalloc((void **)&p, want * sizeof(T));
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};
template <class T>
struct hbuf {                         /* pinned host (or plain malloc for one-shot uploads) */
    T *p = nullptr;
    size_t cap = 0;
    bool pinned = true;
    hipError_t reserve(size_t need)
    {
        if (cap >= need) return hipSuccess;
        release();
        const size_t want = need + need / 4 + 16;
        if (pinned) {
            hipError_t e = hipHostMalloc((void **)&p, want * sizeof(T), hipHostMallocPortable);
            if (e != hipSuccess) { p = nullptr; return e; }
        } else {
            p = (T *)malloc(want * sizeof(T));
            if (!p) return hipErrorOutOfMemory;
        }
        cap = want;
        return hipSuccess;
    }
    void release()
    {
        if (p) { if (pinned) (void)hipHostFree(p); else free(p); }
        p = nullptr;
        cap = 0;
    }
};

/* staging of one chunk (streaming slot) or of one resident batch */
struct stage_t {
    hbuf<uint8_t> h_raw;              /* gather target; unused when the caller's memory is registered */
    hbuf<bsw_dtask> h_tasks;
    hbuf<bsw_rawoff> h_roff;
    hbuf<bsw_result> h_out;
    hbuf<bsw_refx> h_desc;          /* ref mode: target coordinates per seed */
    hbuf<bsw_wireoff> h_woff;
    dbuf<uint8_t> d_raw;
    dbuf<uint64_t> d_seq;
    dbuf<bsw_dtask> d_tasks;
    dbuf<bsw_rawoff> d_roff;
    dbuf<uint32_t> d_order, d_bins;
    dbuf<uint8_t> d_nflag;            /* per seed: bit 0 / 1 = the left / right query holds an N (bsw_pack_kernel -> bsw_bin_count) */
    dbuf<uint64_t> d_keys;            /* per seed: its three list indices (bsw_bin_count -> bsw_bin_scatter) */
    dbuf<bsw_result> d_out;
    dbuf<bsw_pair> d_pair;            /* BSW_RESULT_PAIR: the dense 32-byte records that cross PCIe */
    hbuf<uint64_t> h_blob;            /* small batches: packed sequences | task records | order lists + counters, one DMA */
    dbuf<uint64_t> d_blob;
    dbuf<bsw_refx> d_desc;
    dbuf<bsw_wireoff> d_woff;
    hbuf<uint32_t> h_wout;            /* wire format: the group's 16 KiB result batches as the device wrote them */
    dbuf<uint32_t> d_wout;
    void set_pinned(bool on) { h_raw.pinned = h_tasks.pinned = h_roff.pinned = h_out.pinned = h_desc.pinned = h_woff.pinned = h_blob.pinned = h_wout.pinned = on; }
    void release_host() { h_raw.release(); h_tasks.release(); h_roff.release(); h_out.release(); h_desc.release(); h_woff.release(); h_blob.release(); h_wout.release(); }
    void release_transient_dev() { d_raw.release(); d_roff.release(); d_bins.release(); d_nflag.release(); d_keys.release(); d_desc.release(); d_woff.release(); d_blob.release(); d_wout.release(); }
    void release()
    {
        release_host();
        release_transient_dev();
        d_seq.release(); d_tasks.release(); d_order.release(); d_out.release(); d_pair.release();
    }
};

/* The lane classes of one side are independent launches: they run side by side on auxiliary streams so that one class's
 * tail (its last waves running alone) fills with the other's waves — a 72-column wave (168 registers, 47 KB of LDS per
 * four waves) and a 136-column wave (256, 70 KB) fit one SIMD / one CU together.  One set per slot stream, created right
 * behind it (the runtime deals streams onto its hardware queues in creation order; streams that share a queue run
 * their kernels one after the other, profiles/r3/e2e_hw_queues.txt). */
#define BSW_FORK_AUX (BSW_MAX_LANE_CLASSES - 1)
struct fork_t {
    hipStream_t aux[BSW_FORK_AUX] = {nullptr};
    hipEvent_t ev_nlist = nullptr;    /* behind the N list's launch on a borrowed stream (bsw_binparams.nsplit; forked off at ev_fork_r) */
    hipEvent_t ev_fork = nullptr, ev_fork_r = nullptr, ev_left[BSW_MAX_LANE_CLASSES] = {nullptr}, ev_right[BSW_MAX_LANE_CLASSES] = {nullptr};
    hipEvent_t ev_link[2 * BSW_MAX_LANE_CLASSES] = {nullptr};    /* mode 2: one per link of the chain */
    /* mode 2 (tail fill): a chunk's lane launches form a chain, each released when EVERY workgroup of the one before it
     * has started — flag(i), a device word counted up by launch i's kernel, polled by a sleeping wave in front of launch i+1 */
    uint32_t *flag_mem = nullptr;     /* 2 * BSW_MAX_LANE_CLASSES words, one per 256-byte line */
    uint32_t *flag(int i) const { return flag_mem + 64 * i; }
    uint32_t *expired() const { return flag_mem + 64 * 2 * BSW_MAX_LANE_CLASSES; }   /* waits that ended at their deadline */
    int mode = 0;                     /* 0 none, 1 all classes of a side at once (BSW_FORK=1), 2 tail fill */
    int naux = BSW_FORK_AUX;          /* streams in aux[] (mode 2: the device's other slot streams, borrowed) */
    bool ok = false;
};

struct dev_state {
    int device = 0;
    /* where the card sits: PCI address, NUMA node, and the CPUs next to it that this process may use (empty: not known, or
     * bsw_config.pin_threads = -1) — the slot threads of the device run there and first-touch their pinned staging there */
    std::string bdf;
    int numa_node = -1;
    cpu_set_t cpus;
    int n_cpus = 0;
    std::vector<hipStream_t> streams;
    std::vector<fork_t> forks;        /* one per stream */
    std::vector<hipEvent_t> events;   /* one per stream, for the watchdog */
    std::vector<hipEvent_t> h2d_done; /* one per stream: the chunk's input DMAs have finished */
    std::vector<stage_t> slots;
};

/* Input DMAs of one device run in chunk order, one chunk at a time: chunk k+1's transfer then overlaps chunk k's
 * kernels instead of every slot transferring (and then computing) at the same moment — the TBB fill order of the
 * reference's batch manager (batch_manager.v:418,745-773). */
struct h2d_gate {
    std::mutex mu;
    std::condition_variable cv;
    size_t next = 0;                  /* sequence number of the chunk whose turn it is */
    hipEvent_t last = nullptr;        /* recorded after the previous chunk's input DMAs */
};

struct refbatch_req {
    const uint32_t *in;
    uint32_t *out;
    uint32_t nt;                      /* the batch's task count AS VALIDATED AT SUBMIT (<= BSW_REFBATCH_MAX_TASKS): the header is the
                                         caller's memory until the wait, and nothing downstream re-reads the count from it */
};

struct pipeline;
struct bsw_ctx {
    bsw_config cfg{};
    std::vector<dev_state> devs;
    std::atomic<bool> dead{false};    /* a wait for the GPU timed out: every later call fails fast */
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    bool timed = false;
    /* per-run event pairs since the last bsw_run_history() call (kernel time of every bsw_run) */
    struct run_ev { hipEvent_t e0 = nullptr, mid = nullptr, e1 = nullptr; bool staged = false; };   /* mid: behind pack + bin (bsw_run_staged) */
    std::vector<run_ev> hist;
    size_t hist_used = 0;
    hipEvent_t ev_last0 = nullptr, ev_last1 = nullptr;
    errs err;
    /* async submits: persistent slot threads behind a chunk queue (bsw_batch.hip), started by the first submit */
    struct pipeline *pipe = nullptr;
    /* small synchronous batches (bsw_extend_batch, scalar ABI, wire format) */
    stage_t small;
    /* banded global alignment (F4) */
    dbuf<bsw_gdtask> g_tasks;
    dbuf<uint8_t> g_z;
    dbuf<uint32_t> g_cig, g_order;
    dbuf<bsw_gresult> g_res;
    dbuf<bsw_adtask> a_tasks;         /* local alignment (bsw_align_batch) */
    dbuf<unsigned long long> a_bl;
    dbuf<bsw_kswr> a_res;
    std::vector<refbatch_req> ref_queue;
    int device0() const { return devs[0].device; }
    hipStream_t stream0() const { return devs[0].streams[0]; }
};

struct chunk_info;
struct bsw_dev_batch {
    uint64_t n = 0;
    bsw_dparams P{};
    int variant = 0;
    stage_t st;                       /* device buffers of the batch (host side released after upload) */
    uint64_t seq_words = 0;
    batch_plan plan;
    uint64_t launches = 0;
    uint64_t h2d_bytes = 0;           /* bytes the upload moved over PCIe */
    /* bsw_upload_raw: the bytes as they crossed PCIe, their offsets and the bin scratch stay in HBM, so that bsw_run_staged can
     * redo the device side of the batch manager (pack + bin) in front of the DP kernels */
    bool staged = false;
    chunk_info *ci = nullptr;         /* (chunk_info is declared below; owned) */
    const bsw_ref *ref = nullptr;
};

/* ---- bsw_ctx.hip ---- */
BSW_LOCAL void pin_this_thread(const dev_state &d);       /* the calling thread onto the device's CPUs (no-op when unknown) */
BSW_LOCAL int parse_cpulist(const char *text, cpu_set_t *out);
BSW_LOCAL int wait_event(bsw_ctx *ctx, errs &e, hipEvent_t ev);
BSW_LOCAL int sync_stream(bsw_ctx *ctx, errs &e, hipStream_t st, hipEvent_t ev);
BSW_LOCAL bool is_registered(const void *p, size_t len);
BSW_LOCAL int check_params(errs &e, const bsw_params *p, bsw_dparams *dp);
BSW_LOCAL int mat_max(const int8_t *mat);
BSW_LOCAL int gap_limit(const bsw_params *p, int mx, int qlen, int end_bonus);
BSW_LOCAL size_t nwords(int len);
BSW_LOCAL int busy_check(bsw_ctx *ctx, const char *what);
/* ---- bsw_batch.hip: the streaming pipeline ---- */
BSW_LOCAL bool pipeline_busy(bsw_ctx *ctx);               /* some submit has not been waited for */
BSW_LOCAL void pipeline_shutdown(bsw_ctx *ctx);           /* waits for what is in flight, joins the slot threads */

/* ---- bsw_batch.hip ---- */
/* How one batch is cut into kernel launches lives in batch_plan (above); a chunk's host pass leaves this: */
struct chunk_info {
    size_t words = 0;                 /* seq words the chunk needs */
    const uint8_t *lo = nullptr, *hi = nullptr;   /* span of every sequence the chunk references */
    size_t sum_len = 0;               /* bytes referenced (= gather size) */
    bool direct = false;              /* raw bytes are DMA'd straight out of registered memory */
    bool rev_left = false;            /* left queries sit forwards in raw, their offsets point at the last base (bsw_submit_ref) */
    uint32_t raw_bias = 0;            /* direct: rawoff holds the low 32 bits of the host pointers, raw byte = off - bias */
    bool packed = false;              /* the caller's sequences are 4-bit packed words already (bsw_submit_packed): they are
                                         DMA'd straight into `seq`, no pack kernel */
    batch_plan plan;
    bsw_binparams bp;
    bool streaming = false;           /* (set by the caller) one of MANY chunks of a submit: the slots keep the GPU full, so the chunk is bound by
                                         throughput whatever its size — the lane kernels then keep a list per side (no fused launch) */
    bool rb_on = false;               /* the records' word offsets are still relative: bsw_rebase_kernel runs behind their DMA */
    bsw_rebase rb{};
};

struct gate_turn {                    /* this chunk's place in its device's input-DMA order */
    h2d_gate *gate = nullptr;
    size_t seq = 0;
    hipEvent_t ev = nullptr;
    std::atomic<int> *abort_flag = nullptr;   /* the chunk's submit has failed elsewhere: pass the turn on, do nothing */
    bool *passed = nullptr;           /* set once the turn has been passed on */
};
BSW_LOCAL size_t order_capacity(size_t n);
/* AUTO policy of a chunk once its seeds are counted: lane bins (128 seeds per wavefront), the group kernel (16 per wavefront;
 * returns true — the 16-bit seeds then leave the lane lists for the general kernel: cw16 = their count per wave class, n16
 * their number) or no lane launches at all (bp.lane_on = 0).  group_ok: the scoring parameters allow the packed kernels. */
/* A mid-sized chunk with two-sided seeds as ONE launch: every 8-bit lane seed goes on the left lists (n8 of them; cl / cr: the counted
 * lists), the 8-bit right lists are emptied, bp.fused is set.  kern: bsw_config.kernel; group: the chunk runs the group kernel (else the lane kernels;
 * packed_ok: the scoring parameters allow the packed kernels).  Returns the class for the launch, -1: not fused (lists untouched). */
BSW_LOCAL int fuse_lists(bsw_binparams &bp, int kern, bool group, bool packed_ok, uint32_t n8, uint32_t *cl, uint32_t *cr, bool streaming = false);
BSW_LOCAL void plan_fused(batch_plan &pl, const bsw_binparams &bp, int fused_cls, bool group, const uint32_t *cl);
/* bsw_binparams.nsplit for a chunk (policy), and the N list's place behind the redo list once pl.redo_off / pl.order_len are set
 * (order_len grows by the list; call before bp's offsets are copied from the plan) */
BSW_LOCAL bool nsplit_pays(int kern, const bsw_binparams &bp, bool packed_ok, uint32_t n8, bool streaming, double n_bases = -1.0 /* query bases of the seeds with an N in a query, estimated from a sample; < 0: unknown */);
BSW_LOCAL bool nsplit_candidate(int kern, const bsw_binparams &bp, bool packed_ok, uint32_t n8, bool streaming);       /* worth taking the sample */
BSW_LOCAL void plan_nsplit(batch_plan &pl, bsw_binparams &bp, bool nsplit, uint32_t n_lane);
BSW_LOCAL bool decide_lane_mode(int kern, bool group_ok, bsw_binparams &bp, uint32_t &n_lane, uint32_t n16, uint32_t *cl, uint32_t *cr,
                                uint32_t *cw, const uint32_t *cw16, uint8_t *dep, uint64_t work8_l, uint64_t work8_r, bool streaming = false);
BSW_LOCAL int fill_binparams(errs &e, const bsw_params *p, int kern, bsw_binparams &bp);
BSW_LOCAL bool narrow_foldable(const bsw_binparams &bp);
BSW_LOCAL void narrow_fold(bsw_binparams &bp, uint32_t *cl, uint32_t *cr, uint8_t *dep);
BSW_LOCAL const fork_t *fork_for(const bsw_ctx *ctx, hipStream_t s, bool pipeline = false);
BSW_LOCAL int enqueue_batch(errs &e, const bsw_dparams &P, int variant, const uint64_t *d_seq, const bsw_dtask *d_tasks,
                            uint32_t *d_order, const batch_plan &pl, bsw_result *d_out, hipStream_t s, uint64_t *launches,
                            const fork_t *fk = nullptr, bsw_pair *d_pair = nullptr);
BSW_LOCAL int run_chunk(bsw_ctx *ctx, errs &e, stage_t &st, hipStream_t s, hipEvent_t ev, const bsw_params &p, const bsw_dparams &dp,
                        const bsw_task *tasks, size_t n, bsw_result *out, int gather_threads, const gate_turn *turn = nullptr, bool packed = false);

/* ---- bsw_scalar.hip: the queue the drop-in scalar entry points share ---- */
struct scalar_req {
    int kind = 0;                     /* 0 ksw_extend2, 1 ksw_align2, 2 ksw_global2: all three share the queue and the trip */
    bsw_params p;
    bsw_ext_task t;                   /* extend */
    bsw_ext x;
    bsw_atask at;                     /* align */
    bsw_kswr ar;
    bsw_gtask gt;                     /* global */
    bsw_gresult gr;
    int cap = 0;                      /* CIGAR words this call can take (0: score only) */
    std::vector<uint32_t> cg;
    int rc = 0;
    bool done = false;
};
BSW_LOCAL void scalar_call(scalar_req &req);

#endif
