/*
 * bsw_api.hip — host side of libbwasw_mi355.so: context, batch manager, C ABI.
 *
 * Plays the role of the reference's batch_manager.v + tbb.v + rbb.v (CSR/DSM handshake,
 * 256 KiB task batches in, 16 KiB result batches out, round-robin over 4 PE arrays:
 * batch_manager.v:358-739) on top of the HIP runtime.  The host only validates lengths, counts
 * seeds per kernel class and starts DMAs; packing (byte-per-base -> 16 bases per uint64) and
 * binning (the (qlen, tlen, band) bins of BASELINE.json) run on the GPU (bsw_stage_kernel.hip),
 * chunk k of a submit goes to device k mod n_devices, and the kernels write results in task order.
 * There is no CPU compute path here: every DP cell is evaluated by the HIP kernels.
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "bsw_device.h"
#include "bsw_stage.h"

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
static bool lane_bins_pay(uint32_t n_lane, const uint32_t *cl, const uint32_t *cr)
{
    uint32_t l = 0, r = 0;
    for (int c = 0; c < BSW_MAX_LANE_CLASSES; ++c) { l += cl[c]; r += cr[c]; }
    const uint32_t sides = (l ? 1u : 0u) + (r ? 1u : 0u);
    return n_lane >= (uint32_t)LANE_AUTO_MIN * (sides ? sides : 1u);
}
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

static int fail(errs &e, int code, const char *fmt, ...)
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
    dbuf<bsw_result> d_out;
    dbuf<bsw_pair> d_pair;            /* BSW_RESULT_PAIR: the dense 32-byte records that cross PCIe */
    hbuf<uint64_t> h_blob;            /* small batches: packed sequences | task records | order lists + counters, one DMA */
    dbuf<uint64_t> d_blob;
    dbuf<bsw_refx> d_desc;
    dbuf<bsw_wireoff> d_woff;
    void set_pinned(bool on) { h_raw.pinned = h_tasks.pinned = h_roff.pinned = h_out.pinned = h_desc.pinned = h_woff.pinned = h_blob.pinned = on; }
    void release_host() { h_raw.release(); h_tasks.release(); h_roff.release(); h_out.release(); h_desc.release(); h_woff.release(); h_blob.release(); }
    void release_transient_dev() { d_raw.release(); d_roff.release(); d_bins.release(); d_desc.release(); d_woff.release(); d_blob.release(); }
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
    hipEvent_t ev_fork = nullptr, ev_left[BSW_MAX_LANE_CLASSES] = {nullptr}, ev_right[BSW_MAX_LANE_CLASSES] = {nullptr};
    bool ok = false;
};

struct dev_state {
    int device = 0;
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
};

struct bsw_ctx {
    bsw_config cfg{};
    std::vector<dev_state> devs;
    std::atomic<bool> dead{false};    /* a wait for the GPU timed out: every later call fails fast */
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    bool timed = false;
    /* per-run event pairs since the last bsw_run_history() call (kernel time of every bsw_run) */
    std::vector<std::pair<hipEvent_t, hipEvent_t>> hist;
    size_t hist_used = 0;
    hipEvent_t ev_last0 = nullptr, ev_last1 = nullptr;
    errs err;
    /* async submit */
    std::thread worker;
    bool worker_active = false;
    int worker_rc = 0;
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

struct bsw_dev_batch {
    uint64_t n = 0;
    bsw_dparams P{};
    int variant = 0;
    stage_t st;                       /* device buffers of the batch (host side released after upload) */
    uint64_t seq_words = 0;
    batch_plan plan;
    uint64_t launches = 0;
    uint64_t h2d_bytes = 0;           /* bytes the upload moved over PCIe */
};

/* ---- watchdog: never block in the runtime without a deadline (the RTL documents an
 * inactivity timeout, bwa_mem_sw.v:84-101, but a wedged PE array leaves its busy bit set forever) ---- */
static int wait_event(bsw_ctx *ctx, errs &e, hipEvent_t ev);
static int sync_stream(bsw_ctx *ctx, errs &e, hipStream_t st, hipEvent_t ev)
{
    if (ctx->dead) return fail(e, BSW_E_HIP, "context is dead (an earlier wait for the GPU timed out)");
    HIPCHK(e, hipEventRecord(ev, st));
    return wait_event(ctx, e, ev);
}

static int wait_event(bsw_ctx *ctx, errs &e, hipEvent_t ev)
{
    if (ctx->dead) return fail(e, BSW_E_HIP, "context is dead (an earlier wait for the GPU timed out)");
    const auto t0 = std::chrono::steady_clock::now();
    const double limit = ctx->cfg.timeout_ms > 0 ? (double)ctx->cfg.timeout_ms : 120000.0;
    for (;;) {
        const hipError_t q = hipEventQuery(ev);
        if (q == hipSuccess) return BSW_OK;
        if (q != hipErrorNotReady) return fail(e, BSW_E_HIP, "hipEventQuery: %s", hipGetErrorString(q));
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (ms > limit) {
            ctx->dead = true;
            return fail(e, BSW_E_HIP, "timeout: the GPU did not finish within %d ms; context marked dead", (int)limit);
        }
        /* the first 300 us: spin — a sleep of any length costs ~55 us here (timer slack), a third of a scalar-ABI round
         * trip (profiles/r3/scalar_call_timeline.txt).  After that poll gently: the runtime serialises queries against the
         * other slots' enqueues */
        if (ms < 0.3) { for (int k = 0; k < 64; ++k) __builtin_ia32_pause(); }
        else std::this_thread::sleep_for(std::chrono::microseconds(25));
    }
}

/* ---- registered (DMA-able) host memory ---------------------------------------- */
struct reg_range {
    const uint8_t *lo;
    size_t len;
    bool owned;                       /* from bsw_host_alloc */
};
static std::mutex g_reg_mu;
static std::vector<reg_range> g_regs;

static bool is_registered(const void *p, size_t len)
{
    if (!p) return false;
    std::lock_guard<std::mutex> lk(g_reg_mu);
    const uint8_t *b = (const uint8_t *)p;
    for (const reg_range &r : g_regs)
        if (b >= r.lo && b + len <= r.lo + r.len) return true;
    return false;
}

extern "C" void *bsw_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_reg_mu);
    g_regs.push_back(reg_range{(const uint8_t *)p, bytes, true});
    return p;
}

static int reg_remove(void *p, bool owned)
{
    std::lock_guard<std::mutex> lk(g_reg_mu);
    for (size_t i = 0; i < g_regs.size(); ++i)
        if (g_regs[i].lo == (const uint8_t *)p && g_regs[i].owned == owned) {
            g_regs.erase(g_regs.begin() + (long)i);
            return BSW_OK;
        }
    return BSW_E_INVAL;
}

extern "C" void bsw_host_free(void *p)
{
    if (!p) return;
    if (reg_remove(p, true) == BSW_OK) (void)hipHostFree(p);
}

extern "C" int bsw_host_register(void *p, size_t bytes)
{
    if (!p || bytes == 0) return BSW_E_INVAL;
    const hipError_t e = hipHostRegister(p, bytes, hipHostRegisterPortable);
    if (e != hipSuccess) return e == hipErrorNoDevice ? BSW_E_NODEVICE : BSW_E_HIP;
    std::lock_guard<std::mutex> lk(g_reg_mu);
    g_regs.push_back(reg_range{(const uint8_t *)p, bytes, false});
    return BSW_OK;
}

extern "C" int bsw_host_unregister(void *p)
{
    if (!p) return BSW_E_INVAL;
    if (reg_remove(p, false) != BSW_OK) return BSW_E_INVAL;
    return hipHostUnregister(p) == hipSuccess ? BSW_OK : BSW_E_HIP;
}

/* ------------------------------------------------------------------------- */
extern "C" void bsw_default_params(bsw_params *p)
{
    memset(p, 0, sizeof(*p));
    for (int i = 0; i < 5; ++i)
        for (int j = 0; j < 5; ++j) p->mat[i * 5 + j] = (i == 4 || j == 4) ? -1 : (i == j ? 1 : -4);
    p->o_del = p->o_ins = 6;
    p->e_del = p->e_ins = 1;
    p->w = 100;
    p->pen_clip5 = p->pen_clip3 = 5;
    p->zdrop = 100;
    p->max_band_try = 2;
    p->variant = BSW_VARIANT_H;
}

extern "C" void bsw_default_config(bsw_config *c)
{
    memset(c, 0, sizeof(*c));
    c->device = 0;
    c->kernel = BSW_KERNEL_AUTO;
    c->streams = 4;
    c->pack_threads = 4;
    c->chunk_tasks = 131072;
    c->n_devices = 0;
    c->timeout_ms = 120000;
    c->result_format = BSW_RESULT_FULL;
}

extern "C" int bsw_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int ok = 0;
    for (int d = 0; d < n; ++d) {
        hipDeviceProp_t pr;
        if (hipGetDeviceProperties(&pr, d) == hipSuccess && strncmp(pr.gcnArchName, "gfx950", 6) == 0) ++ok;
    }
    return ok;
}

extern "C" const char *bsw_last_error(const bsw_ctx *ctx) { return ctx ? ctx->err.msg.c_str() : "null ctx"; }

static void ctx_release(bsw_ctx *ctx)
{
    const bool dead = ctx->dead;
    for (auto &d : ctx->devs) {
        (void)hipSetDevice(d.device);
        if (!dead) {
            for (auto s : d.streams) (void)hipStreamSynchronize(s);
            for (auto &f : d.forks) {
                for (auto a : f.aux) if (a) { (void)hipStreamSynchronize(a); (void)hipStreamDestroy(a); }
                if (f.ev_fork) (void)hipEventDestroy(f.ev_fork);
                for (auto ev : f.ev_left) if (ev) (void)hipEventDestroy(ev);
                for (auto ev : f.ev_right) if (ev) (void)hipEventDestroy(ev);
            }
            for (auto s : d.streams) (void)hipStreamDestroy(s);
            for (auto ev : d.events) (void)hipEventDestroy(ev);
            for (auto ev : d.h2d_done) (void)hipEventDestroy(ev);
            for (auto &sl : d.slots) sl.release();
        }
    }
    if (!ctx->devs.empty()) (void)hipSetDevice(ctx->device0());
    if (!dead) {
        if (ctx->ev_start) (void)hipEventDestroy(ctx->ev_start);
        if (ctx->ev_stop) (void)hipEventDestroy(ctx->ev_stop);
        for (auto &pr : ctx->hist) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
        ctx->small.release();
        ctx->g_tasks.release(); ctx->g_z.release(); ctx->g_cig.release(); ctx->g_order.release(); ctx->g_res.release();
        ctx->a_tasks.release(); ctx->a_bl.release(); ctx->a_res.release();
    }
    delete ctx;
}

extern "C" int bsw_create(const bsw_config *cfg, bsw_ctx **out)
{
    if (!out) return BSW_E_INVAL;
    *out = nullptr;
    bsw_config c;
    if (cfg) c = *cfg; else bsw_default_config(&c);
    if (c.streams < 1) c.streams = 2;
    if (c.streams > 8) c.streams = 8;
    if (c.pack_threads < 1) c.pack_threads = 1;
    if (c.chunk_tasks == 0) c.chunk_tasks = 131072;
    if (c.timeout_ms <= 0) c.timeout_ms = 120000;
    if (const char *t = getenv("BSW_TIMEOUT_MS")) { if (atoi(t) > 0) c.timeout_ms = atoi(t); }
    if (c.n_devices < 0 || c.n_devices > BSW_MAX_DEVICES) return BSW_E_INVAL;
    if (c.result_format != BSW_RESULT_FULL && c.result_format != BSW_RESULT_PAIR) return BSW_E_INVAL;
    if (c.n_devices == 0) { c.n_devices = 1; c.devices[0] = c.device; }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        fprintf(stderr, "libbwasw_mi355: no HIP device visible — this library has no CPU path\n");
        return BSW_E_NODEVICE;
    }
    for (int k = 0; k < c.n_devices; ++k) {
        const int dv = c.devices[k];
        if (dv < 0 || dv >= n) return BSW_E_INVAL;
        hipDeviceProp_t pr;
        if (hipGetDeviceProperties(&pr, dv) != hipSuccess) return BSW_E_HIP;
        if (strncmp(pr.gcnArchName, "gfx950", 6) != 0) {
            fprintf(stderr, "libbwasw_mi355: device %d is %s, kernels are built for gfx950 only\n", dv, pr.gcnArchName);
            return BSW_E_NODEVICE;
        }
    }
    c.device = c.devices[0];
    bsw_ctx *ctx = new bsw_ctx();
    ctx->cfg = c;
    ctx->devs.resize((size_t)c.n_devices);
    for (int k = 0; k < c.n_devices; ++k) {
        dev_state &d = ctx->devs[(size_t)k];
        d.device = c.devices[k];
        if (hipSetDevice(d.device) != hipSuccess) { ctx_release(ctx); return BSW_E_HIP; }
        d.slots.resize((size_t)c.streams);
        for (int s = 0; s < c.streams; ++s) {
            hipStream_t st = nullptr;
            hipEvent_t ev = nullptr;
            if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { ctx_release(ctx); return BSW_E_HIP; }
            d.streams.push_back(st);
            {
                /* off unless BSW_FORK=1: with the narrow class folded wherever wider sides exist, the only workload with two
                 * classes per side is 250 bp (136 + 232 columns) — 1 971 GCUPS forked, 1 974 not (gpurun_out/r4h, r4b) */
                static const bool nofork = getenv("BSW_FORK") == nullptr;
                fork_t f;
                bool good = !nofork;
                /* the auxiliary streams run at the LOWEST priority: the widest class of a side (the slot stream's) has the
                 * longest waves and must get its slots first — released at the same instant, the narrow class's many short
                 * workgroups took half the slots and the long waves started late (right side 2.8 ms instead of 2.0,
                 * gpurun_out/r4c trace) */
                int least = 0, greatest = 0;
                static const bool noprio = getenv("BSW_FORK_NOPRIO") != nullptr;
                if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess || noprio) least = 0;
                for (int a = 0; a < BSW_FORK_AUX && good; ++a) good = hipStreamCreateWithPriority(&f.aux[a], hipStreamNonBlocking, least) == hipSuccess;
                good = good && hipEventCreateWithFlags(&f.ev_fork, hipEventDisableTiming) == hipSuccess;
                for (int c = 0; c < BSW_MAX_LANE_CLASSES && good; ++c)
                    good = hipEventCreateWithFlags(&f.ev_left[c], hipEventDisableTiming) == hipSuccess &&
                           hipEventCreateWithFlags(&f.ev_right[c], hipEventDisableTiming) == hipSuccess;
                f.ok = good;
                d.forks.push_back(f);              /* (not ok: the classes of a side run one after the other on the slot stream) */
            }
            if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { ctx_release(ctx); return BSW_E_HIP; }
            d.events.push_back(ev);
            if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { ctx_release(ctx); return BSW_E_HIP; }
            d.h2d_done.push_back(ev);
        }
    }
    if (hipSetDevice(ctx->device0()) != hipSuccess ||
        hipEventCreate(&ctx->ev_start) != hipSuccess || hipEventCreate(&ctx->ev_stop) != hipSuccess) { ctx_release(ctx); return BSW_E_HIP; }
    *out = ctx;
    return BSW_OK;
}

extern "C" void bsw_destroy(bsw_ctx *ctx)
{
    if (!ctx) return;
    if (ctx->worker_active && ctx->worker.joinable()) ctx->worker.join();
    ctx_release(ctx);
}

/* ---- validation ---------------------------------------------------------------- */
static int check_params(errs &e, const bsw_params *p, bsw_dparams *dp)
{
    if (!p) return fail(e, BSW_E_INVAL, "params is NULL");
    if (p->e_del < 1 || p->e_ins < 1 || p->o_del < 0 || p->o_ins < 0)
        return fail(e, BSW_E_INVAL, "need e_del,e_ins >= 1 and o_del,o_ins >= 0");
    if (p->w < 0 || p->w > (1 << 20) || p->max_band_try > 8) return fail(e, BSW_E_INVAL, "band out of range");
    if (p->variant != BSW_VARIANT_H && p->variant != BSW_VARIANT_M) return fail(e, BSW_E_INVAL, "bad variant");
    if (p->o_del + p->e_del > 4096 || p->o_ins + p->e_ins > 4096) return fail(e, BSW_E_LIMIT, "gap penalties too large");
    memset(dp, 0, sizeof(*dp));
    memcpy(dp->mat, p->mat, 25);
    dp->o_del = p->o_del; dp->e_del = p->e_del; dp->o_ins = p->o_ins; dp->e_ins = p->e_ins;
    dp->w = p->w; dp->pen_clip5 = p->pen_clip5; dp->pen_clip3 = p->pen_clip3; dp->zdrop = p->zdrop;
    dp->max_band_try = p->max_band_try > 0 ? p->max_band_try : 1;
    return BSW_OK;
}

static inline int mat_max(const int8_t *mat)
{
    int mx = 0;                                  /* bwa starts the scan at 0 */
    for (int i = 0; i < 25; ++i) mx = mx > mat[i] ? mx : mat[i];
    return mx;
}

/* min(max_ins, max_del): the longest useful gap (sw_pe_array_proc_element.v:925,933 H5/H6);
 * integer form of (int)((double)(qlen*max+end_bonus-o)/e + 1.) */
static inline int gap_limit(const bsw_params *p, int mx, int qlen, int end_bonus)
{
    int mi = (qlen * mx + end_bonus - p->o_ins + p->e_ins) / p->e_ins;
    int md = (qlen * mx + end_bonus - p->o_del + p->e_del) / p->e_del;
    if (mi < 1) mi = 1;
    if (md < 1) md = 1;
    int l = mi < md ? mi : md;
    return l > 65535 ? 65535 : l;
}

static inline size_t nwords(int len) { return (size_t)((len + 15) >> 4); }

/* ---- bsw_pack_bases: the device sequence format, on the host (tools and tests; the batch path packs on the GPU) ---- */
static inline uint64_t squeeze8(uint64_t x)
{
    x = (x | (x >> 4)) & 0x00FF00FF00FF00FFull;
    x = (x | (x >> 8)) & 0x0000FFFF0000FFFFull;
    x = (x | (x >> 16)) & 0x00000000FFFFFFFFull;
    return x;
}

extern "C" int bsw_pack_bases(const uint8_t *s, int len, uint64_t *dst)
{
    if (len < 0 || (len > 0 && (!s || !dst))) return BSW_E_INVAL;
    const int full = len >> 4;
    uint64_t any = 0;
    for (int w = 0; w < full; ++w) {
        uint64_t lo, hi;
        memcpy(&lo, s + 16 * w, 8);
        memcpy(&hi, s + 16 * w + 8, 8);
        if (((lo | hi) & 0xF8F8F8F8F8F8F8F8ull) != 0) {          /* a code > 7: clamp bytewise (never produced by bwa) */
            uint64_t v = 0;
            for (int k = 0; k < 16; ++k) {
                const uint64_t b = s[16 * w + k] > 4 ? 4 : s[16 * w + k];
                v |= b << (k * 4);
            }
            dst[w] = v;
            any |= 4;
            continue;
        }
        /* codes 5..7 -> 4 (N): bit2 set means N, clear the low two bits of such bytes */
        uint64_t nl = lo & 0x0404040404040404ull, nh = hi & 0x0404040404040404ull;
        lo &= ~((nl >> 1) | (nl >> 2));
        hi &= ~((nh >> 1) | (nh >> 2));
        any |= nl | nh;
        dst[w] = squeeze8(lo) | (squeeze8(hi) << 32);
    }
    if (len & 15) {
        uint64_t v = 0;
        for (int k = full * 16; k < len; ++k) {
            const uint64_t b = s[k] > 4 ? 4 : s[k];
            any |= b & 4;
            v |= b << ((k & 15) * 4);
        }
        dst[full] = v;
    }
    return any != 0;
}

/* a whole task array into one packed arena (the caller-side half of bsw_submit_packed) */
extern "C" size_t bsw_pack_tasks_bound(const bsw_task *tasks, size_t n)
{
    size_t w = 0;
    for (size_t i = 0; i < n; ++i) {
        const bsw_task &t = tasks[i];
        if (t.lqlen > 0) w += nwords(t.lqlen) + nwords(t.ltlen > 0 ? t.ltlen : 0);
        if (t.rqlen > 0) w += nwords(t.rqlen) + nwords(t.rtlen > 0 ? t.rtlen : 0);
    }
    return 8 * w + 8;
}

extern "C" int64_t bsw_pack_tasks(const bsw_task *tasks, size_t n, uint64_t *arena, size_t cap, bsw_task *out)
{
    if ((n && (!tasks || !out)) || !arena || ((uintptr_t)arena & 7)) return BSW_E_INVAL;
    size_t w = 0;
    const size_t capw = cap / 8;
    for (size_t i = 0; i < n; ++i) {
        bsw_task t = tasks[i];
        const uint8_t **ptr[4] = {&t.lquery, &t.ltarget, &t.rquery, &t.rtarget};
        const int len[4] = {t.lqlen > 0 ? t.lqlen : 0, t.lqlen > 0 && t.ltlen > 0 ? t.ltlen : 0, t.rqlen > 0 ? t.rqlen : 0, t.rqlen > 0 && t.rtlen > 0 ? t.rtlen : 0};
        for (int k = 0; k < 4; ++k) {
            const size_t nw = nwords(len[k]);
            if (w + nw > capw) return BSW_E_NOMEM;
            if (len[k] && bsw_pack_bases(*ptr[k], len[k], arena + w) < 0) return BSW_E_INVAL;
            *ptr[k] = len[k] ? (const uint8_t *)(arena + w) : nullptr;
            w += nw;
        }
        out[i] = t;
    }
    return (int64_t)(8 * w);
}

/* lane kernel needs a bwa-style matrix (bwa_fill_scmat): a on the diagonal, one mismatch score off it,
 * one score for every pair that involves an N */
static bool lane_matrix_ok(const bsw_params *p)
{
    const int a = p->mat[0], nb = p->mat[1], nn = p->mat[24];
    if (a <= 0 || nb > 0 || nn > a) return false;
    for (int i = 0; i < 5; ++i)
        for (int j = 0; j < 5; ++j)
            if (p->mat[i * 5 + j] != ((i == 4 || j == 4) ? nn : (i == j ? a : nb))) return false;
    return true;
}

static size_t order_capacity(size_t n) { return 4 * n + 16; }   /* upper bound of plan.order_len + the 2 + BSW_MAX_WAVE_CLASSES counters behind it */

/* ---- host pass over a chunk: validate, lay out, count per class ------------------------------ */
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
};

static int fill_binparams(errs &e, const bsw_params *p, int kern, bsw_binparams &bp)
{
    memset(&bp, 0, sizeof(bp));
    bp.a = p->mat[0];
    bp.b = p->mat[1] < 0 ? -p->mat[1] : 0;
    bp.n_wave = bsw::wave_class_count();
    bp.n_lane = bsw::lane_class_count();
    if (bp.n_wave > BSW_MAX_WAVE_CLASSES || bp.n_lane > BSW_MAX_LANE_CLASSES) return fail(e, BSW_E_LIMIT, "class table too large");
    for (int c = 0; c < bp.n_wave; ++c) bp.wave_cols[c] = bsw::wave_class_cols(c);
    for (int c = 0; c < bp.n_lane; ++c) {
        bp.lane_cols[c] = bsw::lane_class_cols(c);
        bp.lane_bits[c] = bsw::lane_class_bits(c);
        if (bp.lane_cols[c] > BSW_LANE_QBINS) return fail(e, BSW_E_LIMIT, "lane class %d has %d columns (> %d)", c, bp.lane_cols[c], BSW_LANE_QBINS);
        if (bp.lane_bits[c] == 8) bp.cols8 = std::max(bp.cols8, bp.lane_cols[c]);
        else bp.cols16 = std::max(bp.cols16, bp.lane_cols[c]);
    }
    bp.lane_on = kern != BSW_KERNEL_WAVE && lane_matrix_ok(p);
    return BSW_OK;
}

/* The narrow lane class (72 columns, three waves per SIMD) is its own launch per side.  It pays where a chunk holds NO
 * wider 8-bit sides — reads with long exact seeds, whose flanks are all short: the 72 / 64 / 40 / 16-column single bins run
 * 12 / 13 / 16 / 24 % faster there (gpurun_out/r4b, r4d).  Beside wider sides it LOSES, whatever their share: one launch
 * per class means the short waves no longer fill the tail of the long ones (longest-first order inside ONE launch is what
 * packs a side onto the wave slots; every launch has a floor of one wave's whole lifetime), and a 72-column wave next to a
 * 136-column one gets no third wave (256 + 168 registers).  Synthetic PE mixed bins (42 % of the lane work in short
 * sides): 2 350 -> 2 080 GCUPS with the launches one after the other, 2 190 on forked streams with priorities; still
 * -15 % with 97 % of the work in short sides (gpurun_out/r4c, r4h, r4i).  So the host decides per chunk: the class is
 * used when its sides hold at least NARROW_MIN_SHARE of the chunk's 8-bit lane work (work of a side ~ its query length:
 * rows ~ 2 qlen, live band ~ constant), otherwise it is folded into the next class (lane_cols = 0: the device's
 * bsw_side_lane_class skips it).  BSW_NARROW_SHARE overrides the threshold (0: always, 2: never). */
#define NARROW_MIN_SHARE 1.0
static bool narrow_foldable(const bsw_binparams &bp)
{
    return bp.n_lane >= 2 && bp.lane_bits[0] == bp.lane_bits[1] && bp.lane_cols[0] > 0 && bp.lane_cols[0] < bp.lane_cols[1];
}
static double narrow_min_share()
{
    static const double v = getenv("BSW_NARROW_SHARE") ? atof(getenv("BSW_NARROW_SHARE")) : NARROW_MIN_SHARE;
    return v;
}
/* fold class 0 into class 1: counts, dependency bits, and the class table the device sorts with */
static void narrow_fold(bsw_binparams &bp, uint32_t *cl, uint32_t *cr, uint8_t *dep)
{
    cl[1] += cl[0]; cr[1] += cr[0]; cl[0] = cr[0] = 0;
    if (dep) {
        for (int lc = 0; lc < BSW_MAX_LANE_CLASSES; ++lc)
            if (dep[lc] & 1u) dep[lc] = (uint8_t)((dep[lc] & ~1u) | 2u);
        dep[1] |= dep[0];
        dep[0] = 0;
    }
    bp.lane_cols[0] = 0;
}

/* tasks[0..n) -> dt[0..n) (device task records), ro[0..n) (gather layout of the raw bytes), class counts -> plan */
/* One pass over the seeds of a chunk: validate, lay the 4-bit arena out, count the kernel classes.  `src(i, tmp, rc)`
 * hands out seed i as a bsw_task (a pointer into the caller's array, or `tmp` filled on the fly — bsw_submit_ref never
 * materialises its tasks); NULL = error rc.  rawoff gets the low 32 bits of every host pointer: when the chunk goes
 * out by direct DMA the pack kernel subtracts raw_bias, otherwise gather_offsets() replaces them. */
template <class Src>
static int prepare_chunk_t(errs &e, const bsw_params *p, int kern, Src &&src, size_t n, bool dev_targets,
                           bsw_dtask *dt, bsw_rawoff *ro, chunk_info &ci, bool rev_left, bool packed = false)
{
    ci.packed = packed;
    const int mx = mat_max(p->mat);
    int rc = fill_binparams(e, p, kern, ci.bp);
    if (rc) return rc;
    bsw_binparams &bp = ci.bp;
    uint64_t acc = 0, accb = 0;
    const uint8_t *lo = (const uint8_t *)UINTPTR_MAX, *hi = nullptr;
    uint32_t cw_all[BSW_MAX_WAVE_CLASSES] = {0}, cw[BSW_MAX_WAVE_CLASSES] = {0};
    uint32_t cl[BSW_MAX_LANE_CLASSES] = {0}, cr[BSW_MAX_LANE_CLASSES] = {0}, n_lane = 0;
    uint8_t dep[BSW_MAX_LANE_CLASSES] = {0};
    uint64_t lane_work[BSW_MAX_LANE_CLASSES] = {0};          /* sum of query lengths per lane class (narrow_fold) */
    auto span = [&](const uint8_t *s, int len) {
        if (len > 0) { if (s < lo) lo = s; if (s + len > hi) hi = s + len; }
    };
    bsw_task tmp;
    /* H5/H6 per query length, computed once per length and chunk (two integer divisions each: with them per seed they were
     * most of this pass) */
    std::vector<uint16_t> gl5((size_t)BSW_MAX_QLEN + 1, 0), gl3((size_t)BSW_MAX_QLEN + 1, 0);
    const auto glim = [&](std::vector<uint16_t> &tab, int qlen, int clip) -> uint16_t {
        uint16_t &v = tab[(size_t)qlen];
        if (!v) v = (uint16_t)gap_limit(p, mx, qlen, clip);       /* >= 1: zero means not computed yet */
        return v;
    };
    /* packed input: whether the words can be DMA'd as they lie (registered, compact arena) decides the word offsets, and
     * the staging records are write-combined memory that must not be read back — so the arena span is found first */
    bool packed_direct = false;
    const uint8_t *plo = nullptr;
    if (packed) {
        const uint8_t *l0 = (const uint8_t *)UINTPTR_MAX, *h0 = nullptr;
        uint64_t sum = 0;
        auto sp = [&](const uint8_t *s, int len) {
            if (len > 0 && s) { const size_t nb = 8 * nwords(len); sum += nb; if (s < l0) l0 = s; if (s + nb > h0) h0 = s + nb; }
        };
        for (size_t i = 0; i < n; ++i) {
            const bsw_task *tp = src(i, tmp, rc);
            if (!tp) return rc;
            if (tp->lqlen > 0) { sp(tp->lquery, tp->lqlen); sp(tp->ltarget, tp->ltlen); }
            if (tp->rqlen > 0) { sp(tp->rquery, tp->rqlen); sp(tp->rtarget, tp->rtlen); }
        }
        const size_t spb = h0 ? (size_t)(h0 - l0) : 0;
        packed_direct = spb > 0 && spb < (1ull << 32) - RAW_SLACK && spb <= 2 * sum + (1u << 20) && is_registered(l0, spb);
        plo = l0;
    }
    for (size_t i = 0; i < n; ++i) {
        const bsw_task *tp = src(i, tmp, rc);
        if (!tp) return rc;
        const bsw_task &t = *tp;
        if (t.lqlen < 0 || t.rqlen < 0 || t.ltlen < 0 || t.rtlen < 0)
            return fail(e, BSW_E_INVAL, "task %zu: negative length", i);
        if (t.lqlen > BSW_MAX_QLEN || t.rqlen > BSW_MAX_QLEN || t.ltlen > BSW_MAX_TLEN || t.rtlen > BSW_MAX_TLEN)
            return fail(e, BSW_E_LIMIT, "task %zu: length beyond BSW_MAX_QLEN/BSW_MAX_TLEN", i);
        if (t.h0 <= 0) return fail(e, BSW_E_INVAL, "task %zu: h0 must be > 0", i);
        if ((int64_t)t.h0 + (int64_t)(t.lqlen + t.rqlen) * mx >= BSW_MAX_SCORE)
            return fail(e, BSW_E_LIMIT, "task %zu: score range beyond BSW_MAX_SCORE", i);
        if ((t.lqlen && (!t.lquery || (t.ltlen && !t.ltarget && !dev_targets))) ||
            (t.rqlen && (!t.rquery || (t.rtlen && !t.rtarget && !dev_targets))))
            return fail(e, BSW_E_INVAL, "task %zu: NULL sequence pointer", i);
        if (t.wlim_l < 0 || t.wlim_r < 0) return fail(e, BSW_E_INVAL, "task %zu: negative wlim", i);
        bsw_dtask d;                                /* built here, stored once: dt[] is write-combined staging */
        bsw_rawoff r;
        memset(&d, 0, sizeof(d));
        memset(&r, 0, sizeof(r));
        if (packed) {
            /* 16 bases per uint64 already (base k in bits [4k, 4k+3], codes 0-3 = ACGT, 4-7 = N), every sequence on an
             * 8-byte boundary: the spans are whole words, rawoff keeps the pointers' low bits until the arena base is known */
            if ((t.lqlen && (((uintptr_t)t.lquery | (t.ltlen ? (uintptr_t)t.ltarget : 0)) & 7)) ||
                (t.rqlen && (((uintptr_t)t.rquery | (t.rtlen ? (uintptr_t)t.rtarget : 0)) & 7)))
                return fail(e, BSW_E_INVAL, "task %zu: packed sequences must start on 8-byte boundaries", i);
            /* direct: the registered arena IS the device's seq buffer, word offsets relative to its lowest word (an empty
             * target takes its query's offset: word 0 of a target may be read even when no row is) */
            if (t.lqlen) {
                d.lq_off = packed_direct ? (uint32_t)((t.lquery - plo) >> 3) : (uint32_t)acc;
                acc += nwords(t.lqlen);
                d.lt_off = packed_direct ? (uint32_t)(((t.ltlen ? t.ltarget : t.lquery) - plo) >> 3) : (uint32_t)acc;
                acc += nwords(t.ltlen);
                accb += 8ull * (nwords(t.lqlen) + nwords(t.ltlen));
                span(t.lquery, 8 * (int)nwords(t.lqlen)); span(t.ltarget, 8 * (int)nwords(t.ltlen));
            }
            if (t.rqlen) {
                d.rq_off = packed_direct ? (uint32_t)((t.rquery - plo) >> 3) : (uint32_t)acc;
                acc += nwords(t.rqlen);
                d.rt_off = packed_direct ? (uint32_t)(((t.rtlen ? t.rtarget : t.rquery) - plo) >> 3) : (uint32_t)acc;
                acc += nwords(t.rtlen);
                accb += 8ull * (nwords(t.rqlen) + nwords(t.rtlen));
                span(t.rquery, 8 * (int)nwords(t.rqlen)); span(t.rtarget, 8 * (int)nwords(t.rtlen));
            }
        } else {
        if (t.lqlen) {
            d.lq_off = (uint32_t)acc; acc += nwords(t.lqlen);
            d.lt_off = (uint32_t)acc; acc += nwords(t.ltlen);
            r.lq = (uint32_t)(uintptr_t)t.lquery; accb += (uint64_t)t.lqlen;
            span(rev_left ? t.lquery - (t.lqlen - 1) : t.lquery, t.lqlen);    /* rev_left: lquery points at the LAST base, read backwards */
            if (!dev_targets) { r.lt = (uint32_t)(uintptr_t)t.ltarget; accb += (uint64_t)t.ltlen; span(t.ltarget, t.ltlen); }
        }
        if (t.rqlen) {
            d.rq_off = (uint32_t)acc; acc += nwords(t.rqlen);
            d.rt_off = (uint32_t)acc; acc += nwords(t.rtlen);
            r.rq = (uint32_t)(uintptr_t)t.rquery; accb += (uint64_t)t.rqlen;
            span(t.rquery, t.rqlen);
            if (!dev_targets) { r.rt = (uint32_t)(uintptr_t)t.rtarget; accb += (uint64_t)t.rtlen; span(t.rtarget, t.rtlen); }
        }
        }
        d.lqlen = (uint16_t)t.lqlen; d.rqlen = (uint16_t)t.rqlen;
        d.ltlen = (uint16_t)t.ltlen; d.rtlen = (uint16_t)t.rtlen;
        /* H5/H6: host-supplied band limits win over the library's formula (proc_element.v:925,933) */
        d.wlim_l = (uint16_t)(t.wlim_l > 0 ? std::min(t.wlim_l, 65535) : glim(gl5, t.lqlen, p->pen_clip5));
        d.wlim_r = (uint16_t)(t.wlim_r > 0 ? std::min(t.wlim_r, 65535) : glim(gl3, t.rqlen, p->pen_clip3));
        d.h0 = t.h0; d.init_score = t.init_score; d.qbeg = t.qbeg; d.tag = t.tag;
        dt[i] = d;
        if (!packed) ro[i] = r;                     /* (packed input has no byte offsets) */
        /* class counts (the device sorts with the same functions) */
        const int qm = t.lqlen > t.rqlen ? t.lqlen : t.rqlen;
        const int wc = bsw_wave_class_of(&bp, qm);
        if (wc < 0) return fail(e, BSW_E_LIMIT, "task %zu: no kernel class", i);
        ++cw_all[wc];
        const int bits = bsw_seed_lane_bits(&bp, t.lqlen, t.rqlen, t.h0);
        if (!bits) ++cw[wc];
        else {
            ++n_lane;
            int lc = -1;
            if (t.lqlen) {
                const int c = lc = bsw_side_lane_class(&bp, bits, t.lqlen);
                if (c < 0) return fail(e, BSW_E_LIMIT, "task %zu: no lane class", i);
                ++cl[c];
                lane_work[c] += (uint64_t)t.lqlen;
            }
            if (t.rqlen) {
                const int c = bsw_side_lane_class(&bp, bits, t.rqlen);
                if (c < 0) return fail(e, BSW_E_LIMIT, "task %zu: no lane class", i);
                ++cr[c];
                lane_work[c] += (uint64_t)t.rqlen;
                if (lc >= 0) dep[lc] |= (uint8_t)(1u << c);
            }
        }
    }
    if (acc >= (1ull << 32)) return fail(e, BSW_E_LIMIT, "batch sequence arena beyond 2^32 words; split the batch");
    if (accb >= (1ull << 32) - RAW_SLACK) return fail(e, BSW_E_LIMIT, "batch holds more than 4 GiB of bases; split the batch");
    if (kern == BSW_KERNEL_AUTO && !lane_bins_pay(n_lane, cl, cr)) bp.lane_on = 0;
    if (bp.lane_on && narrow_foldable(bp)) {
        uint64_t all8 = 0;
        for (int c = 0; c < bp.n_lane; ++c) if (bp.lane_bits[c] == bp.lane_bits[0]) all8 += lane_work[c];
        if ((double)lane_work[0] < narrow_min_share() * (double)all8) narrow_fold(bp, cl, cr, dep);
    }
    if (!bp.lane_on) {
        memcpy(cw, cw_all, sizeof(cw));
        memset(cl, 0, sizeof(cl));
        memset(cr, 0, sizeof(cr));
        n_lane = 0;
    }
    batch_plan &pl = ci.plan;
    pl = batch_plan();
    for (int c = 0; c < BSW_MAX_WAVE_CLASSES; ++c) pl.wave_start[c + 1] = pl.wave_start[c] + cw[c];
    uint32_t cur = pl.wave_start[BSW_MAX_WAVE_CLASSES];
    pl.lane_all_off = cur;
    pl.lane_all_cnt = n_lane;
    cur += n_lane;
    for (int c = 0; c < BSW_MAX_LANE_CLASSES; ++c) { pl.laneL_off[c] = cur; cur += cl[c]; }
    pl.laneL_off[BSW_MAX_LANE_CLASSES] = cur;
    for (int c = 0; c < BSW_MAX_LANE_CLASSES; ++c) { pl.laneR_off[c] = cur; cur += cr[c]; }
    pl.laneR_off[BSW_MAX_LANE_CLASSES] = cur;
    pl.redo_off = cur;
    pl.order_len = cur + n_lane;
    pl.redo_cls = bsw_wave_class_of(&bp, std::max(bp.cols8, bp.cols16) - 1);
    memcpy(pl.dep, dep, sizeof(pl.dep));
    memcpy(bp.wave_start, pl.wave_start, sizeof(bp.wave_start));
    bp.lane_all_off = pl.lane_all_off;
    memcpy(bp.laneL_off, pl.laneL_off, sizeof(bp.laneL_off));
    memcpy(bp.laneR_off, pl.laneR_off, sizeof(bp.laneR_off));
    ci.words = (size_t)acc;
    ci.sum_len = (size_t)accb;
    ci.lo = hi ? lo : nullptr;
    ci.hi = hi;
    /* DMA the caller's arena as it is when it is registered memory and not much larger than what it holds */
    const size_t spanb = hi ? (size_t)(hi - lo) : 0;
    ci.direct = spanb > 0 && spanb < (1ull << 32) - RAW_SLACK && spanb <= 2 * ci.sum_len + (1u << 20) && is_registered(lo, spanb);
    ci.rev_left = rev_left && ci.direct;            /* the gather path mirrors the left queries while copying */
    ci.raw_bias = ci.direct ? (uint32_t)(uintptr_t)lo : 0u;
    {
        static const bool dbg = getenv("BSW_DEBUG_TIMING") != nullptr;
        if (dbg) fprintf(stderr, "[bsw] chunk n=%zu: %s%s, span %zu B for %zu B referenced\n", n, packed ? "packed " : "", ci.direct ? "direct DMA" : "gather", spanb, ci.sum_len);
    }
    if (packed) {
        ci.direct = packed_direct;
        if (packed_direct) ci.words = spanb >> 3;
    }
    return BSW_OK;
}

/* the gather path: rawoff = where gather_raw puts each sequence in the pinned staging arena (back to back) */
static void gather_offsets(const bsw_task *tasks, size_t n, bool dev_targets, bsw_rawoff *ro)
{
    uint64_t accb = 0;
    for (size_t i = 0; i < n; ++i) {
        const bsw_task &t = tasks[i];
        bsw_rawoff &r = ro[i];
        memset(&r, 0, sizeof(r));
        if (t.lqlen) {
            r.lq = (uint32_t)accb; accb += (uint64_t)t.lqlen;
            if (!dev_targets) { r.lt = (uint32_t)accb; accb += (uint64_t)t.ltlen; }
        }
        if (t.rqlen) {
            r.rq = (uint32_t)accb; accb += (uint64_t)t.rqlen;
            if (!dev_targets) { r.rt = (uint32_t)accb; accb += (uint64_t)t.rtlen; }
        }
    }
}

static int prepare_chunk(errs &e, const bsw_params *p, int kern, const bsw_task *tasks, size_t n, bool dev_targets,
                         bsw_dtask *dt, bsw_rawoff *ro, chunk_info &ci, bool rev_left = false, bool packed = false)
{
    int rc = prepare_chunk_t(e, p, kern, [tasks](size_t i, bsw_task &, int &) { return tasks + i; }, n, dev_targets, dt, ro, ci, rev_left, packed);
    if (!rc && !ci.direct && !packed) gather_offsets(tasks, n, dev_targets, ro);
    return rc;
}

/* packed sequences that are not in registered memory: their words go to the pinned staging arena in seq layout */
static void gather_packed(const bsw_task *tasks, const bsw_dtask *dt, size_t n, uint64_t *dst)
{
    for (size_t i = 0; i < n; ++i) {
        const bsw_task &t = tasks[i];
        const bsw_dtask &d = dt[i];
        if (t.lqlen) {
            memcpy(dst + d.lq_off, t.lquery, 8 * nwords(t.lqlen));
            if (t.ltlen) memcpy(dst + d.lt_off, t.ltarget, 8 * nwords(t.ltlen));
        }
        if (t.rqlen) {
            memcpy(dst + d.rq_off, t.rquery, 8 * nwords(t.rqlen));
            if (t.rtlen) memcpy(dst + d.rt_off, t.rtarget, 8 * nwords(t.rtlen));
        }
    }
}

/* copy the sequences of tasks[0..n) into the pinned staging arena laid out by prepare_chunk */
static void gather_raw(const bsw_task *tasks, const bsw_rawoff *ro, size_t n, bool dev_targets, uint8_t *dst, int threads, bool rev_left = false)
{
    auto work = [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
            const bsw_task &t = tasks[i];
            const bsw_rawoff &r = ro[i];
            if (t.lqlen) {
                if (rev_left) for (int k = 0; k < t.lqlen; ++k) dst[r.lq + (uint32_t)k] = *(t.lquery - k);
                else memcpy(dst + r.lq, t.lquery, (size_t)t.lqlen);
                if (!dev_targets && t.ltlen) memcpy(dst + r.lt, t.ltarget, (size_t)t.ltlen);
            }
            if (t.rqlen) {
                memcpy(dst + r.rq, t.rquery, (size_t)t.rqlen);
                if (!dev_targets && t.rtlen) memcpy(dst + r.rt, t.rtarget, (size_t)t.rtlen);
            }
        }
    };
    if (threads <= 1 || n < 4096) { work(0, n); return; }
    std::vector<std::thread> th;
    const size_t per = (n + (size_t)threads - 1) / (size_t)threads;
    for (int k = 1; k < threads; ++k) {
        const size_t lo = per * (size_t)k, hi = std::min(n, lo + per);
        if (lo < hi) th.emplace_back(work, lo, hi);
    }
    work(0, std::min(n, per));
    for (auto &t : th) t.join();
}

/* ---- device side of a chunk: DMA, pack, (fetch), bin, and optionally the DP kernels ---------- */
static const fork_t *fork_for(const bsw_ctx *ctx, hipStream_t s)
{
    for (const dev_state &d : ctx->devs)
        for (size_t k = 0; k < d.streams.size(); ++k)
            if (d.streams[k] == s) return k < d.forks.size() && d.forks[k].ok ? &d.forks[k] : nullptr;
    return nullptr;
}

static int enqueue_batch(errs &e, const bsw_dparams &P, int variant, const uint64_t *d_seq, const bsw_dtask *d_tasks,
                         uint32_t *d_order, const batch_plan &pl, bsw_result *d_out, hipStream_t s, uint64_t *launches,
                         const fork_t *fk = nullptr, bsw_pair *d_pair = nullptr)
{
    const int nc = bsw::wave_class_count();
    /* device words behind the order lists: [0] the redo list's length, [1 + c] the work counter of wave class c's launch,
     * [1 + BSW_MAX_WAVE_CLASSES] the redo launch's — zeroed here, on the stream, before anything counts in them */
    uint32_t *redo_cnt = d_order + pl.order_len;
    HIPCHK(e, hipMemsetAsync(redo_cnt, 0, (2 + BSW_MAX_WAVE_CLASSES) * sizeof(uint32_t), s));
    for (int c = 0; c < nc; ++c) {
        const uint32_t cnt = pl.wave_start[c + 1] - pl.wave_start[c];
        if (!cnt) continue;
        HIPCHK(e, bsw::launch_wave(c, variant, P, d_seq, d_tasks, d_order + pl.wave_start[c], cnt, nullptr, redo_cnt + 1 + c, d_out, s));
        if (d_pair) HIPCHK(e, bsw::launch_pairs_from_results(d_order + pl.wave_start[c], cnt, nullptr, d_out, d_pair, s));
        if (launches) ++*launches;
    }
    if (pl.lane_all_cnt) {
        const int nlc = bsw::lane_class_count();
        /* The classes of a side side by side: the k-th non-empty class of a side (widest first: its waves run longest) goes
         * to stream k — the slot stream, then the auxiliary ones.  A right-side launch waits for exactly the left-side
         * launches that hold one of its seeds (plan.dep); streams are in-order, so only other streams' launches need an event. */
        hipStream_t lstream[BSW_MAX_LANE_CLASSES] = {nullptr}, rstream[BSW_MAX_LANE_CLASSES] = {nullptr};
        int nl = 0, nr = 0;
        for (int c = nlc - 1; c >= 0; --c) {
            if (pl.laneL_off[c + 1] - pl.laneL_off[c]) { lstream[c] = (fk && nl > 0 && nl <= BSW_FORK_AUX) ? fk->aux[nl - 1] : s; ++nl; }
            if (pl.laneR_off[c + 1] - pl.laneR_off[c]) { rstream[c] = (fk && nr > 0 && nr <= BSW_FORK_AUX) ? fk->aux[nr - 1] : s; ++nr; }
        }
        const bool forked = fk && (nl > 1 || nr > 1);
        if (forked) {
            HIPCHK(e, hipEventRecord(fk->ev_fork, s));                 /* everything queued on s so far (input DMAs, pack, bins) */
            for (int a = 0; a < BSW_FORK_AUX; ++a) HIPCHK(e, hipStreamWaitEvent(fk->aux[a], fk->ev_fork, 0));
        }
        for (int c = nlc - 1; c >= 0; --c) {
            const uint32_t cnt = pl.laneL_off[c + 1] - pl.laneL_off[c];
            if (!cnt) continue;
            HIPCHK(e, bsw::launch_lane(c, variant, P, 0, d_seq, d_tasks, d_order + pl.laneL_off[c], cnt, d_out, lstream[c]));
            if (forked) HIPCHK(e, hipEventRecord(fk->ev_left[c], lstream[c]));
            if (launches) ++*launches;
        }
        for (int c = nlc - 1; c >= 0; --c) {
            const uint32_t cnt = pl.laneR_off[c + 1] - pl.laneR_off[c];
            if (!cnt) continue;
            if (forked)
                for (int lc = 0; lc < nlc; ++lc)
                    if (lstream[lc] && lstream[lc] != rstream[c] && ((pl.dep[lc] >> c) & 1)) HIPCHK(e, hipStreamWaitEvent(rstream[c], fk->ev_left[lc], 0));
            HIPCHK(e, bsw::launch_lane(c, variant, P, 1, d_seq, d_tasks, d_order + pl.laneR_off[c], cnt, d_out, rstream[c]));
            if (forked && rstream[c] != s) HIPCHK(e, hipEventRecord(fk->ev_right[c], rstream[c]));
            if (launches) ++*launches;
        }
        if (forked) {                                                   /* join: the slot stream waits for whatever ran elsewhere */
            for (int c = 0; c < nlc; ++c) {
                if (lstream[c] && lstream[c] != s) HIPCHK(e, hipStreamWaitEvent(s, fk->ev_left[c], 0));
                if (rstream[c] && rstream[c] != s) HIPCHK(e, hipStreamWaitEvent(s, fk->ev_right[c], 0));
            }
        }
        HIPCHK(e, bsw::launch_finalize(P, d_tasks, d_order + pl.lane_all_off, pl.lane_all_cnt, d_out,
                                       d_order + pl.redo_off, redo_cnt, d_pair, s));
        /* seeds whose first band try was not final: recompute from scratch, one wavefront each */
        HIPCHK(e, bsw::launch_wave(pl.redo_cls, variant, P, d_seq, d_tasks, d_order + pl.redo_off, pl.lane_all_cnt,
                                   redo_cnt, redo_cnt + 1 + BSW_MAX_WAVE_CLASSES, d_out, s));
        if (d_pair) HIPCHK(e, bsw::launch_pairs_from_results(d_order + pl.redo_off, pl.lane_all_cnt, redo_cnt, d_out, d_pair, s));
        if (launches) *launches += 2;
    }
    return BSW_OK;
}

/* the raw bytes and the task records are in st.h_* (or the caller's registered arena): move them, pack, bin */
struct gate_turn {                    /* this chunk's place in its device's input-DMA order */
    h2d_gate *gate = nullptr;
    size_t seq = 0;
    hipEvent_t ev = nullptr;
    std::atomic<int> *abort_flag = nullptr;
};

static int stage_device(errs &e, stage_t &st, hipStream_t s, const chunk_info &ci, size_t n, bool dev_targets,
                        const bsw_ref *ref, uint64_t *h2d_bytes, const gate_turn *turn = nullptr, size_t dev_index = 0)
{
    const size_t n_desc = ref ? n : 0;              /* st.h_desc: one bsw_refx per seed */
    const size_t rawb = ci.packed ? 0 : (ci.direct ? (size_t)(ci.hi - ci.lo) : ci.sum_len);
    hipError_t he;
    if ((he = st.d_raw.reserve(rawb + RAW_FRONT + RAW_SLACK)) != hipSuccess || (he = st.d_seq.reserve(ci.words + 4)) != hipSuccess ||
        (he = st.d_tasks.reserve(n + 1)) != hipSuccess || (he = st.d_roff.reserve(n + 1)) != hipSuccess ||
        (he = st.d_order.reserve(order_capacity(n))) != hipSuccess || (he = st.d_bins.reserve(BSW_BIN_WORDS)) != hipSuccess ||
        (he = st.d_out.reserve(n + 1)) != hipSuccess || (n_desc && (he = st.d_desc.reserve(n_desc)) != hipSuccess))
        return fail(e, BSW_E_NOMEM, "device staging: %s", hipGetErrorString(he));
    {
        std::unique_lock<std::mutex> lk;
        if (turn) {
            lk = std::unique_lock<std::mutex>(turn->gate->mu);
            turn->gate->cv.wait(lk, [&]() { return turn->gate->next == turn->seq || *turn->abort_flag; });
            if (*turn->abort_flag) return fail(e, BSW_E_HIP, "aborted: another chunk failed");
            if (turn->gate->last) HIPCHK(e, hipStreamWaitEvent(s, turn->gate->last, 0));
        }
        hipError_t ce = hipSuccess;
        if (ci.packed) {               /* the words are the device layout already: they land in `seq`, nothing is packed */
            if (ci.words) ce = hipMemcpyAsync(st.d_seq.p, ci.direct ? (const void *)ci.lo : (const void *)st.h_raw.p, ci.words * 8, hipMemcpyHostToDevice, s);
        } else if (rawb) ce = hipMemcpyAsync(st.d_raw.p + RAW_FRONT, ci.direct ? ci.lo : st.h_raw.p, rawb, hipMemcpyHostToDevice, s);
        if (ce == hipSuccess) ce = hipMemcpyAsync(st.d_tasks.p, st.h_tasks.p, n * sizeof(bsw_dtask), hipMemcpyHostToDevice, s);
        if (ce == hipSuccess && !ci.packed) ce = hipMemcpyAsync(st.d_roff.p, st.h_roff.p, n * sizeof(bsw_rawoff), hipMemcpyHostToDevice, s);
        if (ce == hipSuccess && n_desc) ce = hipMemcpyAsync(st.d_desc.p, st.h_desc.p, n_desc * sizeof(bsw_refx), hipMemcpyHostToDevice, s);
        if (turn) {
            if (ce == hipSuccess) ce = hipEventRecord(turn->ev, s);
            if (ce == hipSuccess) turn->gate->last = turn->ev;
            turn->gate->next = turn->seq + 1;          /* pass the turn on even on failure: nobody may wait forever */
            turn->gate->cv.notify_all();
        }
        if (ce != hipSuccess) return fail(e, BSW_E_HIP, "input DMA: %s", hipGetErrorString(ce));
    }
    (void)dev_targets;
    if (!ci.packed)
        HIPCHK(e, bsw::launch_pack(st.d_raw.p + RAW_FRONT, st.d_tasks.p, st.d_roff.p, ci.raw_bias, (uint32_t)n, ci.rev_left ? 1 : 0,
                                   ref ? ref->d_pac[dev_index] : nullptr, ref ? ref->l_pac : 0, ref ? st.d_desc.p : nullptr, st.d_seq.p, s));
    HIPCHK(e, bsw::launch_bin(ci.bp, st.d_seq.p, st.d_tasks.p, (uint32_t)n, st.d_bins.p, st.d_order.p, s));
    if (h2d_bytes) *h2d_bytes = (ci.packed ? ci.words * 8 : rawb + n * sizeof(bsw_rawoff)) + n * sizeof(bsw_dtask) + n_desc * sizeof(bsw_refx);
    return BSW_OK;
}

/* ---- batch plan export (host only) ---------------------------------------------------------- */
static void plan_segments(const batch_plan &pl, uint32_t *seg)
{
    int k = 0;
    for (int c = 0; c < 8; ++c) seg[k++] = pl.wave_start[c];
    seg[k++] = pl.lane_all_off;                    /* 8 */
    for (int c = 0; c < 8; ++c) seg[k++] = pl.laneL_off[std::min(c, BSW_MAX_LANE_CLASSES)];    /* 9..16 */
    for (int c = 0; c < 8; ++c) seg[k++] = pl.laneR_off[std::min(c, BSW_MAX_LANE_CLASSES)];    /* 17..24 */
    seg[k++] = pl.redo_off;                        /* 25 */
    seg[k++] = pl.order_len;                       /* 26 */
}

extern "C" int64_t bsw_plan_batch(const bsw_params *p, const bsw_task *tasks, size_t n, int kernel, int pack_threads,
                                  uint32_t *order, uint32_t *seg)
{
    (void)pack_threads;
    if (!p || (!tasks && n) || !seg) return BSW_E_INVAL;
    errs e;
    bsw_dparams dp;
    int rc = check_params(e, p, &dp);
    if (rc) return rc;
    std::vector<bsw_dtask> dt(n ? n : 1);
    std::vector<bsw_rawoff> ro(n ? n : 1);
    chunk_info ci;
    rc = prepare_chunk(e, p, kernel, tasks, n, false, dt.data(), ro.data(), ci);
    if (rc) return rc;
    plan_segments(ci.plan, seg);
    if (order) {
        /* the device's rules (bsw_bin_count/scan/scatter) replayed on the host: lists by class, lane sides by
         * (class, query with / without an N, query length descending); the order inside one query length is task
         * order here, arbitrary there */
        const bsw_binparams &bp = ci.bp;
        std::vector<uint32_t> cur(BSW_BIN_WORDS, 0), hist(BSW_BIN_WAVE0, 0);
        auto has_n = [](const uint8_t *q, int len) {
            for (int j = 0; j < len; ++j)
                if (q[j] >= 4) return 1;
            return 0;
        };
        auto keys = [&](size_t i, int &k0, int &k1, int &k2) {
            const bsw_dtask &T = dt[i];
            k1 = k2 = -1;
            const int bits = bsw_seed_lane_bits(&bp, T.lqlen, T.rqlen, T.h0);
            if (!bits) { k0 = BSW_BIN_WAVE0 + bsw_wave_class_of(&bp, std::max(T.lqlen, T.rqlen)); return; }
            k0 = BSW_BIN_LANEALL;
            if (T.lqlen) k1 = BSW_BIN_SIDE(0, bsw_side_lane_class(&bp, bits, T.lqlen), has_n(tasks[i].lquery, T.lqlen), T.lqlen);
            if (T.rqlen) k2 = BSW_BIN_SIDE(1, bsw_side_lane_class(&bp, bits, T.rqlen), has_n(tasks[i].rquery, T.rqlen), T.rqlen);
        };
        for (size_t i = 0; i < n; ++i) {
            int k0, k1, k2;
            keys(i, k0, k1, k2);
            if (k1 >= 0) ++hist[(size_t)k1];
            if (k2 >= 0) ++hist[(size_t)k2];
        }
        for (int side = 0; side < 2; ++side)
            for (int c = 0; c < bp.n_lane; ++c) {
                uint32_t run = side ? bp.laneR_off[c] : bp.laneL_off[c];
                for (int hn = 1; hn >= 0; --hn)
                    for (int q = BSW_LANE_QBINS - 1; q >= 0; --q) {
                        const size_t idx = (size_t)BSW_BIN_SIDE(side, c, hn, q);
                        cur[idx] = run;
                        run += hist[idx];
                    }
            }
        for (int c = 0; c < bp.n_wave; ++c) cur[(size_t)(BSW_BIN_WAVE0 + c)] = bp.wave_start[c];
        cur[BSW_BIN_LANEALL] = bp.lane_all_off;
        for (size_t i = 0; i < n; ++i) {
            int k0, k1, k2;
            keys(i, k0, k1, k2);
            order[cur[(size_t)k0]++] = (uint32_t)i;
            if (k1 >= 0) order[cur[(size_t)k1]++] = (uint32_t)i;
            if (k2 >= 0) order[cur[(size_t)k2]++] = (uint32_t)i;
        }
    }
    return (int64_t)ci.words;
}

/* ---- device-resident batches ------------------------------------------------ */
extern "C" void bsw_free_batch(bsw_ctx *ctx, bsw_dev_batch *b)
{
    if (!b) return;
    if (ctx) (void)hipSetDevice(ctx->device0());
    b->st.release();
    delete b;
}

static int busy_check(bsw_ctx *ctx, const char *what)
{
    if (ctx->dead) return fail(ctx->err, BSW_E_HIP, "%s: context is dead (an earlier wait for the GPU timed out)", what);
    if (ctx->worker_active) return fail(ctx->err, BSW_E_BUSY, "%s: a bsw_submit is in flight (call bsw_wait first)", what);
    return BSW_OK;
}

static void fill_refx(const bsw_ref_task *rt, size_t n, bsw_refx *x);

static int upload_common(bsw_ctx *ctx, const bsw_params *p, const bsw_task *tasks, size_t n, bsw_dev_batch **out,
                         const bsw_ref *ref /* NULL: targets come from the host */, const bsw_ref_task *rtasks, bool packed = false)
{
    *out = nullptr;
    errs &e = ctx->err;
    if (n >= (1ull << 32)) return fail(e, BSW_E_LIMIT, "more than 2^32-1 tasks in one batch");
    bsw_dparams dp;
    int rc = check_params(e, p, &dp);
    if (rc) return rc;
    HIPCHK(e, hipSetDevice(ctx->device0()));
    bsw_dev_batch *b = new bsw_dev_batch();
    stage_t &st = b->st;
    st.set_pinned(false);               /* one-shot upload: plain host staging, synchronous copies */
    chunk_info ci;
    if (st.h_tasks.reserve(n + 1) != hipSuccess || st.h_roff.reserve(n + 1) != hipSuccess) { bsw_free_batch(ctx, b); return fail(e, BSW_E_NOMEM, "host staging"); }
    rc = prepare_chunk(e, p, ctx->cfg.kernel, tasks, n, ref != nullptr, st.h_tasks.p, st.h_roff.p, ci, false, packed);
    if (rc) { bsw_free_batch(ctx, b); return rc; }
    if (!ci.direct) {
        if (st.h_raw.reserve(ci.sum_len + RAW_SLACK) != hipSuccess) { bsw_free_batch(ctx, b); return fail(e, BSW_E_NOMEM, "host staging"); }
        if (packed) gather_packed(tasks, st.h_tasks.p, n, (uint64_t *)st.h_raw.p);
        else gather_raw(tasks, st.h_roff.p, n, ref != nullptr, st.h_raw.p, ctx->cfg.pack_threads);
    }
    if (ref) {
        if (st.h_desc.reserve(n + 1) != hipSuccess) { bsw_free_batch(ctx, b); return fail(e, BSW_E_NOMEM, "host staging"); }
        fill_refx(rtasks, n, st.h_desc.p);
    }
    b->n = n; b->P = dp; b->variant = p->variant; b->seq_words = ci.words; b->plan = ci.plan;
    hipStream_t s = ctx->stream0();
    rc = stage_device(e, st, s, ci, n, ref != nullptr, ref, &b->h2d_bytes);
    if (!rc && n) {
        hipError_t he = hipMemsetAsync(st.d_out.p, 0xff, n * sizeof(bsw_result), s);
        if (he != hipSuccess) rc = fail(e, BSW_E_HIP, "memset: %s", hipGetErrorString(he));
    }
    if (!rc) rc = sync_stream(ctx, e, s, ctx->devs[0].events[0]);
    if (rc) { bsw_free_batch(ctx, b); return rc; }
    st.release_host();
    st.release_transient_dev();
    *out = b;
    return BSW_OK;
}

extern "C" int bsw_upload(bsw_ctx *ctx, const bsw_params *p, const bsw_task *tasks, size_t n, bsw_dev_batch **out)
{
    if (!ctx) return BSW_E_INVAL;
    if (!out || (!tasks && n)) return fail(ctx->err, BSW_E_INVAL, "bsw_upload: NULL argument");
    int rc = busy_check(ctx, "bsw_upload");
    if (rc) return rc;
    return upload_common(ctx, p, tasks, n, out, nullptr, nullptr);
}

/* bsw_upload for sequences that are 4-bit packed already (see bsw_submit_packed) */
extern "C" int bsw_upload_packed(bsw_ctx *ctx, const bsw_params *p, const bsw_task *tasks, size_t n, bsw_dev_batch **out)
{
    if (!ctx) return BSW_E_INVAL;
    if (!out || (!tasks && n)) return fail(ctx->err, BSW_E_INVAL, "bsw_upload_packed: NULL argument");
    int rc = busy_check(ctx, "bsw_upload_packed");
    if (rc) return rc;
    return upload_common(ctx, p, tasks, n, out, nullptr, nullptr, true);
}

/* ---- device-resident reference (F3) ------------------------------------------------ */
extern "C" void bsw_ref_free(bsw_ctx *ctx, bsw_ref *ref);

extern "C" int bsw_ref_upload(bsw_ctx *ctx, const uint8_t *pac, int64_t l_pac, bsw_ref **out)
{
    if (!ctx) return BSW_E_INVAL;
    errs &e = ctx->err;
    if (!pac || !out || l_pac <= 0) return fail(e, BSW_E_INVAL, "bsw_ref_upload: bad argument");
    *out = nullptr;
    bsw_ref *r = new bsw_ref();
    r->l_pac = l_pac;
    r->d_pac.assign(ctx->devs.size(), nullptr);
    const size_t bytes = (size_t)((l_pac + 3) >> 2);
    for (size_t d = 0; d < ctx->devs.size(); ++d) {               /* every GPU of the context keeps its own copy */
        hipError_t he = hipSetDevice(ctx->devs[d].device);
        if (he == hipSuccess) he = hipMalloc((void **)&r->d_pac[d], bytes + 8);
        if (he == hipSuccess) he = hipMemcpy(r->d_pac[d], pac, bytes, hipMemcpyHostToDevice);
        if (he != hipSuccess) {
            bsw_ref_free(ctx, r);
            return fail(e, BSW_E_HIP, "pac upload to device %d: %s", ctx->devs[d].device, hipGetErrorString(he));
        }
    }
    (void)hipSetDevice(ctx->device0());
    *out = r;
    return BSW_OK;
}

extern "C" void bsw_ref_free(bsw_ctx *ctx, bsw_ref *ref)
{
    if (!ref) return;
    for (size_t d = 0; d < ref->d_pac.size(); ++d) {
        if (!ref->d_pac[d]) continue;
        if (ctx && d < ctx->devs.size()) (void)hipSetDevice(ctx->devs[d].device);
        (void)hipFree(ref->d_pac[d]);
    }
    if (ctx) (void)hipSetDevice(ctx->device0());
    delete ref;
}

/* mem_chain2aln's task extraction minus the target bases, which stay on the device: one seed of a
 * read -> one task.  rev_left: the left query is NOT copied reversed; lquery points at its last base (query[qbeg-1]). */
static int ref_to_task(errs &e, const bsw_params *p, int64_t l_pac, const bsw_ref_task &r, size_t i, bool rev_left,
                       uint8_t *scratch, size_t &so, bsw_task &t)
{
    const bsw_seed &sd = r.seed;
    const int64_t two = l_pac << 1;
    if (!r.query || r.l_query < 1 || sd.qbeg < 0 || sd.len < 1 || sd.qbeg + sd.len > r.l_query)
        return fail(e, BSW_E_INVAL, "ref task %zu: bad seed / read", i);
    if (r.rmax0 < 0 || r.rmax1 > two || r.rmax0 > sd.rbeg || r.rmax1 < sd.rbeg + sd.len || (r.rmax0 < l_pac && l_pac < r.rmax1))
        return fail(e, BSW_E_INVAL, "ref task %zu: window outside the reference or bridging the strands", i);
    const int64_t lt = sd.rbeg - r.rmax0, rtl = r.rmax1 - (sd.rbeg + sd.len);
    if (lt > BSW_MAX_TLEN || rtl > BSW_MAX_TLEN) return fail(e, BSW_E_LIMIT, "ref task %zu: window beyond BSW_MAX_TLEN", i);
    memset(&t, 0, sizeof(t));
    if (sd.qbeg > 0) {
        if (rev_left) t.lquery = r.query + sd.qbeg - 1;
        else {
            for (int k = 0; k < sd.qbeg; ++k) scratch[so + (size_t)k] = r.query[sd.qbeg - 1 - k];
            t.lquery = scratch + so;
            so += (size_t)sd.qbeg;
        }
        t.lqlen = sd.qbeg; t.ltlen = (int32_t)lt;
    }
    if (sd.qbeg + sd.len != r.l_query) {
        t.rquery = r.query + sd.qbeg + sd.len; t.rqlen = r.l_query - (sd.qbeg + sd.len); t.rtlen = (int32_t)rtl;
    }
    t.h0 = sd.len * p->mat[0]; t.init_score = r.init_score; t.qbeg = sd.qbeg; t.tag = r.tag;
    return BSW_OK;
}

/* where the device finds the two targets of every seed in the resident pac */
static void fill_refx(const bsw_ref_task *rt, size_t n, bsw_refx *x)
{
    for (size_t i = 0; i < n; ++i) x[i] = bsw_refx{rt[i].seed.rbeg - 1, rt[i].seed.rbeg + rt[i].seed.len};
}

extern "C" int bsw_upload_ref(bsw_ctx *ctx, const bsw_params *p, const bsw_ref *ref, const bsw_ref_task *rt, size_t n, bsw_dev_batch **out)
{
    if (!ctx) return BSW_E_INVAL;
    errs &e = ctx->err;
    if (!p || !ref || !out || (!rt && n)) return fail(e, BSW_E_INVAL, "bsw_upload_ref: NULL argument");
    int rc = busy_check(ctx, "bsw_upload_ref");
    if (rc) return rc;
    std::vector<bsw_task> tasks(n ? n : 1);
    size_t scratch_len = 0;
    for (size_t i = 0; i < n; ++i) scratch_len += (size_t)(rt[i].seed.qbeg > 0 ? rt[i].seed.qbeg : 0);
    std::vector<uint8_t> scratch(scratch_len + 1);
    size_t so = 0;
    for (size_t i = 0; i < n; ++i) {
        rc = ref_to_task(e, p, ref->l_pac, rt[i], i, false, scratch.data(), so, tasks[i]);
        if (rc) return rc;
    }
    return upload_common(ctx, p, tasks.data(), n, out, ref, rt);
}

extern "C" int bsw_extend_ref(bsw_ctx *ctx, const bsw_params *p, const bsw_ref *ref, const bsw_ref_task *rt, size_t n, bsw_result *out)
{
    if (!ctx) return BSW_E_INVAL;
    if (!out && n) return fail(ctx->err, BSW_E_INVAL, "bsw_extend_ref: NULL argument");
    bsw_dev_batch *b = nullptr;
    int rc = bsw_upload_ref(ctx, p, ref, rt, n, &b);
    if (rc) return rc;
    rc = bsw_run(ctx, b);
    if (!rc) rc = bsw_download(ctx, b, out);
    bsw_free_batch(ctx, b);
    return rc;
}

extern "C" int bsw_run(bsw_ctx *ctx, bsw_dev_batch *b)
{
    if (!ctx) return BSW_E_INVAL;
    errs &e = ctx->err;
    if (!b) return fail(e, BSW_E_INVAL, "bsw_run: NULL argument");
    int rc = busy_check(ctx, "bsw_run");
    if (rc) return rc;
    HIPCHK(e, hipSetDevice(ctx->device0()));
    hipStream_t s = ctx->stream0();
    hipEvent_t e0 = ctx->ev_start, e1 = ctx->ev_stop;
    if (ctx->hist_used < 4096) {
        if (ctx->hist_used == ctx->hist.size()) {
            hipEvent_t a, c;
            HIPCHK(e, hipEventCreate(&a));
            HIPCHK(e, hipEventCreate(&c));
            ctx->hist.emplace_back(a, c);
        }
        e0 = ctx->hist[ctx->hist_used].first;
        e1 = ctx->hist[ctx->hist_used].second;
        ++ctx->hist_used;
    }
    HIPCHK(e, hipEventRecord(e0, s));
    b->launches = 0;
    rc = enqueue_batch(e, b->P, b->variant, b->st.d_seq.p, b->st.d_tasks.p, b->st.d_order.p, b->plan, b->st.d_out.p, s, &b->launches, fork_for(ctx, s));
    if (rc) return rc;
    HIPCHK(e, hipEventRecord(e1, s));
    ctx->ev_last0 = e0; ctx->ev_last1 = e1;
    ctx->timed = true;
    return BSW_OK;
}

extern "C" int bsw_sync(bsw_ctx *ctx)
{
    if (!ctx) return BSW_E_INVAL;
    errs &e = ctx->err;
    if (ctx->worker_active) return fail(e, BSW_E_BUSY, "bsw_sync: a bsw_submit is in flight (call bsw_wait)");
    HIPCHK(e, hipSetDevice(ctx->device0()));
    return sync_stream(ctx, e, ctx->stream0(), ctx->devs[0].events[0]);
}

extern "C" int bsw_last_run_ms(bsw_ctx *ctx, float *ms)
{
    if (!ctx || !ms || !ctx->timed) return BSW_E_INVAL;
    errs &e = ctx->err;
    int rc = bsw_sync(ctx);
    if (rc) return rc;
    HIPCHK(e, hipEventElapsedTime(ms, ctx->ev_last0, ctx->ev_last1));
    return BSW_OK;
}

extern "C" int bsw_run_history(bsw_ctx *ctx, float *ms, int cap)
{
    if (!ctx || (!ms && cap > 0)) return BSW_E_INVAL;
    errs &e = ctx->err;
    int rc = bsw_sync(ctx);
    if (rc) return rc;
    int n = 0;
    for (size_t i = 0; i < ctx->hist_used && n < cap; ++i, ++n)
        HIPCHK(e, hipEventElapsedTime(&ms[n], ctx->hist[i].first, ctx->hist[i].second));
    ctx->hist_used = 0;
    return n;
}

extern "C" int bsw_download(bsw_ctx *ctx, bsw_dev_batch *b, bsw_result *out)
{
    if (!ctx) return BSW_E_INVAL;
    errs &e = ctx->err;
    if (!b || (!out && b->n)) return fail(e, BSW_E_INVAL, "bsw_download: NULL argument");
    int rc = bsw_sync(ctx);
    if (rc) return rc;
    HIPCHK(e, hipMemcpy(out, b->st.d_out.p, b->n * sizeof(bsw_result), hipMemcpyDeviceToHost));
    return BSW_OK;
}

extern "C" int bsw_batch_info(const bsw_dev_batch *b, uint64_t *n_tasks, uint64_t *in_bytes, uint64_t *out_bytes, uint64_t *n_launches)
{
    if (!b) return BSW_E_INVAL;
    if (n_tasks) *n_tasks = b->n;
    if (in_bytes) *in_bytes = b->seq_words * 8 + b->n * sizeof(bsw_dtask) + (uint64_t)b->plan.redo_off * sizeof(uint32_t);
    if (out_bytes) *out_bytes = b->n * sizeof(bsw_result);
    if (n_launches) *n_launches = b->launches;
    return BSW_OK;
}

extern "C" int bsw_batch_order(bsw_ctx *ctx, const bsw_dev_batch *b, uint32_t *order, uint32_t *seg)
{
    if (!ctx) return BSW_E_INVAL;
    errs &e = ctx->err;
    if (!b || !seg) return fail(e, BSW_E_INVAL, "bsw_batch_order: NULL argument");
    int rc = bsw_sync(ctx);
    if (rc) return rc;
    plan_segments(b->plan, seg);
    if (order && b->plan.redo_off)
        HIPCHK(e, hipMemcpy(order, b->st.d_order.p, (size_t)b->plan.redo_off * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return BSW_OK;
}

/* ---- a HANDFUL of seeds (the scalar ksw_extend2 entry points, tiny batches): no device-side staging at all.
 * The batch path's pack kernel, two memsets and three binning kernels are seven launches of ~6 us each in front of the DP
 * kernel (profiles/r3/scalar_call_timeline.txt: 58 us before the extension starts).  For up to SMALL_BATCH seeds the host
 * packs the bases (a few hundred bytes), sorts the seeds into their general-kernel classes, and ONE DMA carries
 * sequences, task records, order lists and zeroed counters; then the DP kernel(s), then the result copy. ---- */
#define SMALL_BATCH 256
static int run_small(bsw_ctx *ctx, errs &e, stage_t &st, hipStream_t s, hipEvent_t ev, const bsw_params &p, const bsw_dparams &dp,
                     const bsw_task *tasks, size_t n, bsw_result *out)
{
    std::vector<bsw_dtask> dt(n);
    std::vector<bsw_rawoff> ro(n);
    chunk_info ci;
    int rc = prepare_chunk(e, &p, BSW_KERNEL_WAVE, tasks, n, false, dt.data(), ro.data(), ci);   /* validation, word offsets, class counts */
    if (rc) return rc;
    const batch_plan &pl = ci.plan;
    const size_t n_ctr = 2 + BSW_MAX_WAVE_CLASSES;
    const size_t w_seq = ci.words + 4, w_tasks = (n * sizeof(bsw_dtask) + 7) / 8, w_order = ((pl.order_len + n_ctr) * sizeof(uint32_t) + 7) / 8;
    const size_t total = w_seq + w_tasks + w_order;
    hipError_t he;
    if ((he = st.h_blob.reserve(total)) != hipSuccess) return fail(e, BSW_E_NOMEM, "pinned staging: %s", hipGetErrorString(he));
    if ((he = st.d_blob.reserve(total)) != hipSuccess) return fail(e, BSW_E_NOMEM, "device staging: %s", hipGetErrorString(he));
    uint64_t *hb = st.h_blob.p;
    memset(hb, 0, total * sizeof(uint64_t));
    for (size_t i = 0; i < n; ++i) {                 /* the device sequence format, packed here */
        const bsw_task &t = tasks[i];
        const bsw_dtask &d = dt[i];
        if (t.lqlen) { bsw_pack_bases(t.lquery, t.lqlen, hb + d.lq_off); if (t.ltlen) bsw_pack_bases(t.ltarget, t.ltlen, hb + d.lt_off); }
        if (t.rqlen) { bsw_pack_bases(t.rquery, t.rqlen, hb + d.rq_off); if (t.rtlen) bsw_pack_bases(t.rtarget, t.rtlen, hb + d.rt_off); }
    }
    memcpy(hb + w_seq, dt.data(), n * sizeof(bsw_dtask));
    uint32_t *ho = (uint32_t *)(hb + w_seq + w_tasks);
    {
        uint32_t cur[BSW_MAX_WAVE_CLASSES];
        for (int c = 0; c < BSW_MAX_WAVE_CLASSES; ++c) cur[c] = pl.wave_start[c];
        for (size_t i = 0; i < n; ++i) ho[cur[bsw_wave_class_of(&ci.bp, std::max(tasks[i].lqlen, tasks[i].rqlen))]++] = (uint32_t)i;
    }
    HIPCHK(e, hipMemcpyAsync(st.d_blob.p, hb, total * sizeof(uint64_t), hipMemcpyHostToDevice, s));
    const uint64_t *d_seq = st.d_blob.p;
    const bsw_dtask *d_tasks = (const bsw_dtask *)(st.d_blob.p + w_seq);
    uint32_t *d_order = (uint32_t *)(st.d_blob.p + w_seq + w_tasks), *ctr = d_order + pl.order_len;   /* (zero: copied that way) */
    /* the result records go straight to pinned (device-visible) host memory: a few 96-byte stores over the link instead of
     * a device buffer, a copy and its launch */
    const bool out_direct = is_registered(out, n * sizeof(bsw_result));
    if (!out_direct && (he = st.h_out.reserve(n)) != hipSuccess) return fail(e, BSW_E_NOMEM, "pinned staging: %s", hipGetErrorString(he));
    bsw_result *res = out_direct ? out : st.h_out.p;
    const int nc = bsw::wave_class_count();
    for (int c = 0; c < nc; ++c) {
        const uint32_t cnt = pl.wave_start[c + 1] - pl.wave_start[c];
        if (cnt) HIPCHK(e, bsw::launch_wave(c, p.variant, dp, d_seq, d_tasks, d_order + pl.wave_start[c], cnt, nullptr, ctr + 1 + c, res, s));
    }
    rc = sync_stream(ctx, e, s, ev);
    if (rc) return rc;
    if (!out_direct) memcpy(out, st.h_out.p, n * sizeof(bsw_result));
    return BSW_OK;
}

/* ---- one synchronous chunk through a staging slot (small batches; the streaming workers use the same steps) ---- */
static int run_chunk(bsw_ctx *ctx, errs &e, stage_t &st, hipStream_t s, hipEvent_t ev, const bsw_params &p, const bsw_dparams &dp,
                     const bsw_task *tasks, size_t n, bsw_result *out, int gather_threads, const gate_turn *turn = nullptr, bool packed = false)
{
    if (n == 0) return BSW_OK;
    {
        static const bool nosmall = getenv("BSW_NO_SMALL") != nullptr;     /* (measurements) */
        if (!nosmall && n <= SMALL_BATCH && !packed && !turn && ctx->cfg.kernel != BSW_KERNEL_LANE && ctx->cfg.result_format == BSW_RESULT_FULL)
            return run_small(ctx, e, st, s, ev, p, dp, tasks, n, out);
    }
    static const bool dbg = getenv("BSW_DEBUG_TIMING") != nullptr;
    auto tnow = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_a = dbg ? tnow() : 0;
    hipError_t he;
    if ((he = st.h_tasks.reserve(n + 1)) != hipSuccess || (he = st.h_roff.reserve(n + 1)) != hipSuccess)
        return fail(e, BSW_E_NOMEM, "pinned staging: %s", hipGetErrorString(he));
    chunk_info ci;
    int rc = prepare_chunk(e, &p, ctx->cfg.kernel, tasks, n, false, st.h_tasks.p, st.h_roff.p, ci, false, packed);
    if (rc) return rc;
    const double t_b = dbg ? tnow() : 0;
    if (!ci.direct) {
        if ((he = st.h_raw.reserve(ci.sum_len + RAW_SLACK)) != hipSuccess) return fail(e, BSW_E_NOMEM, "pinned staging: %s", hipGetErrorString(he));
        if (packed) gather_packed(tasks, st.h_tasks.p, n, (uint64_t *)st.h_raw.p);
        else gather_raw(tasks, st.h_roff.p, n, false, st.h_raw.p, gather_threads);
    }
    const double t_c = dbg ? tnow() : 0;
    rc = stage_device(e, st, s, ci, n, false, nullptr, nullptr, turn);
    if (rc) return rc;
    /* BSW_RESULT_PAIR: `out` addresses bsw_pair[n]; the dense 32-byte records come from their own device array */
    const bool pairs = ctx->cfg.result_format == BSW_RESULT_PAIR;
    const size_t rec = pairs ? sizeof(bsw_pair) : sizeof(bsw_result);
    if (pairs && (he = st.d_pair.reserve(n + 1)) != hipSuccess) return fail(e, BSW_E_NOMEM, "device staging: %s", hipGetErrorString(he));
    rc = enqueue_batch(e, dp, p.variant, st.d_seq.p, st.d_tasks.p, st.d_order.p, ci.plan, st.d_out.p, s, nullptr, fork_for(ctx, s), pairs ? st.d_pair.p : nullptr);
    if (rc) return rc;
    const bool out_direct = is_registered(out, n * rec);
    if (!out_direct && (he = st.h_out.reserve(n)) != hipSuccess) return fail(e, BSW_E_NOMEM, "pinned staging: %s", hipGetErrorString(he));
    HIPCHK(e, hipMemcpyAsync(out_direct ? (void *)out : (void *)st.h_out.p, pairs ? (const void *)st.d_pair.p : (const void *)st.d_out.p, n * rec, hipMemcpyDeviceToHost, s));
    const double t_d = dbg ? tnow() : 0;
    rc = sync_stream(ctx, e, s, ev);
    if (rc) return rc;
    const double t_e = dbg ? tnow() : 0;
    if (!out_direct) memcpy((void *)out, st.h_out.p, n * rec);
    if (dbg) fprintf(stderr, "[bsw] chunk n=%zu %s: prepare %.3f ms, gather %.3f, enqueue %.3f, gpu wait %.3f, copy-out %.3f (t0=%.3f)\n", n,
                     ci.direct ? "direct" : "gather", t_b - t_a, t_c - t_b, t_d - t_c, t_e - t_d, tnow() - t_e, t_a);
    return BSW_OK;
}

/* ---- streaming submit: one host thread per (device, slot); chunk k -> device k mod G, slot (k / G) mod S —
 * the round-robin of the reference's four TBB/RBB pairs over its PE arrays (batch_manager.v:343-348,418,745-773) ---- */
struct chunk_span {
    size_t base, cnt;
};

/* tasks[0..n) -> per-device chunk lists.  Chunks are equal-sized (a short tail chunk would fall below the lane
 * kernel's minimum batch), chunk c of the plan belongs to device c mod G, and every device's first
 * chunk is cut in two so that its first DMA — the only one no kernel overlaps — is short. */
static std::vector<std::vector<chunk_span>> plan_chunks(size_t n, size_t chunk, size_t G)
{
    std::vector<std::vector<chunk_span>> out(G);
    if (n == 0) return out;
    size_t nch = (n + chunk / 2) / chunk;
    if (nch == 0) nch = 1;
    size_t per = ((n + nch - 1) / nch + 255) & ~(size_t)255;
    size_t c = 0;
    for (size_t base = 0; base < n; base += per, ++c) {
        const size_t cnt = std::min(per, n - base);
        std::vector<chunk_span> &v = out[c % G];
        if (v.empty() && cnt >= 3 * (size_t)LANE_AUTO_MIN + 1024) {     /* (the smaller part still holds a lane launch's worth of one-sided seeds; two-sided ones take the general kernels there, at the same cost) */
            const size_t h = ((cnt / 3) + 255) & ~(size_t)255;
            v.push_back(chunk_span{base, h});
            v.push_back(chunk_span{base + h, cnt - h});
        } else
            v.push_back(chunk_span{base, cnt});
    }
    return out;
}

/* One slot = one host thread + one stream + one set of staging buffers.  Per chunk: host pass (validate, lay out,
 * count) -> wait for the slot's previous chunk -> input DMAs in the device's chunk order -> pack, bin, DP kernels,
 * result DMA.  The host pass of chunk k+S runs while chunk k is still on the GPU: it only needs the pinned host
 * staging, which is free again as soon as chunk k's input DMAs are done. */
static int slot_worker(bsw_ctx *ctx, const bsw_params &p, const bsw_dparams &dp, const bsw_task *tasks,
                       const bsw_ref *ref, const bsw_ref_task *rtasks,   /* non-NULL: seeds against the device-resident reference */
                       bsw_result *out, const std::vector<chunk_span> &chunks, size_t d, size_t s, int gather_threads,
                       std::atomic<int> &abort_flag, h2d_gate &gate, errs &e, bool packed)
{
    std::vector<bsw_task> rt_tasks;                 /* ref mode: this chunk's seeds as tasks (left queries by reference) */
    static const bool dbg = getenv("BSW_DEBUG_TIMING") != nullptr;
    auto tnow = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_host = 0, t_staging = 0, t_finish = 0, t_stage = 0;
    dev_state &dev = ctx->devs[d];
    const size_t S = dev.slots.size();
    stage_t &st = dev.slots[s];
    hipStream_t stream = dev.streams[s];
    struct { bool active = false; size_t n = 0; char *out = nullptr; bool direct = false, copied = false; } pend;
    /* BSW_RESULT_PAIR: `out` addresses bsw_pair[n]; the dense 32-byte records come from their own device array */
    const bool pairs = ctx->cfg.result_format == BSW_RESULT_PAIR;
    const size_t rec = pairs ? sizeof(bsw_pair) : sizeof(bsw_result);
    auto d_res = [&]() -> const void * { return pairs ? (const void *)st.d_pair.p : (const void *)st.d_out.p; };
    bool queued = false;                            /* some async op of the current chunk may be on the stream (set before the first one) */
    auto bail = [&](int rc) {                       /* wake the slots waiting for their DMA turn; leave nothing in flight */
        abort_flag = 1;
        { std::lock_guard<std::mutex> lk(gate.mu); }
        gate.cv.notify_all();
        /* a failure after stage_device queued its first copy leaves DMAs out of the caller's registered arena and kernels
         * in flight although pend.active is still false: drain the stream (with the watchdog) before the error is reported,
         * so the caller may free or reuse that memory as soon as bsw_wait returns */
        if (pend.active || queued) { errs quiet; (void)sync_stream(ctx, quiet, stream, dev.events[s]); pend.active = false; queued = false; }
        return rc;
    };
    hipError_t he = hipSetDevice(dev.device);
    if (he != hipSuccess) return bail(fail(e, BSW_E_HIP, "hipSetDevice: %s", hipGetErrorString(he)));
    auto finish = [&]() -> int {                    /* the slot's chunk in flight: wait (watchdog), hand the results over */
        if (!pend.active) return BSW_OK;
        pend.active = false;
        int rc = BSW_OK;
        if (!pend.copied) {                         /* kernels done -> result DMA -> done */
            rc = wait_event(ctx, e, dev.events[s]);
            if (rc) return rc;
            const hipError_t ce = hipMemcpyAsync(pend.direct ? (void *)pend.out : (void *)st.h_out.p, d_res(), pend.n * rec, hipMemcpyDeviceToHost, stream);
            if (ce != hipSuccess) return fail(e, BSW_E_HIP, "result DMA: %s", hipGetErrorString(ce));
        }
        rc = sync_stream(ctx, e, stream, dev.events[s]);
        if (rc) return rc;
        if (!pend.direct) memcpy(pend.out, st.h_out.p, pend.n * rec);
        return BSW_OK;
    };
    for (size_t k = s; k < chunks.size() && !abort_flag; k += S) {            /* k-th chunk of this device */
        const bsw_task *ct = tasks ? tasks + chunks[k].base : nullptr;
        const size_t n = chunks[k].cnt;
        int rc = BSW_OK;
        const double t0 = dbg ? tnow() : 0;
        if (pend.active) rc = wait_event(ctx, e, dev.h2d_done[s]);           /* pinned host staging is free again */
        if (rc) return bail(rc);
        const double t1 = dbg ? tnow() : 0;
        if ((he = st.h_tasks.reserve(n + 1)) != hipSuccess || (he = st.h_roff.reserve(n + 1)) != hipSuccess ||
            (rtasks && (he = st.h_desc.reserve(n + 1)) != hipSuccess))
            return bail(fail(e, BSW_E_NOMEM, "pinned staging: %s", hipGetErrorString(he)));
        chunk_info ci;
        if (rtasks) {                               /* mem_chain2aln's task extraction fused into the pass; the targets stay on the device */
            const bsw_ref_task *crt = rtasks + chunks[k].base;
            const size_t base = chunks[k].base;
            bsw_refx *rx = st.h_desc.p;
            size_t so = 0;
            rc = prepare_chunk_t(e, &p, ctx->cfg.kernel, [&](size_t i, bsw_task &tmp, int &erc) -> const bsw_task * {
                erc = ref_to_task(e, &p, ref->l_pac, crt[i], base + i, true, nullptr, so, tmp);
                rx[i] = bsw_refx{crt[i].seed.rbeg - 1, crt[i].seed.rbeg + crt[i].seed.len};
                return erc ? nullptr : &tmp;
            }, n, true, st.h_tasks.p, st.h_roff.p, ci, true);
            if (rc) return bail(rc);
            if (!ci.direct) {                       /* reads in pageable memory: materialise the tasks for the gather */
                rt_tasks.resize(n);
                for (size_t i = 0; i < n && !rc; ++i) rc = ref_to_task(e, &p, ref->l_pac, crt[i], base + i, true, nullptr, so, rt_tasks[i]);
                if (rc) return bail(rc);
                ct = rt_tasks.data();
                gather_offsets(ct, n, true, st.h_roff.p);
            }
        } else {
            rc = prepare_chunk(e, &p, ctx->cfg.kernel, ct, n, false, st.h_tasks.p, st.h_roff.p, ci, false, packed);
            if (rc) return bail(rc);
        }
        if (!ci.direct) {
            if ((he = st.h_raw.reserve(ci.sum_len + RAW_SLACK)) != hipSuccess) return bail(fail(e, BSW_E_NOMEM, "pinned staging: %s", hipGetErrorString(he)));
            if (packed) gather_packed(ct, st.h_tasks.p, n, (uint64_t *)st.h_raw.p);
            else gather_raw(ct, st.h_roff.p, n, rtasks != nullptr, st.h_raw.p, gather_threads, rtasks != nullptr);
        }
        const double t2 = dbg ? tnow() : 0;
        rc = finish();
        if (rc) return bail(rc);
        const double t3 = dbg ? tnow() : 0;
        gate_turn turn;
        turn.gate = &gate; turn.seq = k; turn.ev = dev.h2d_done[s]; turn.abort_flag = &abort_flag;
        queued = true;
        rc = stage_device(e, st, stream, ci, n, rtasks != nullptr, ref, nullptr, &turn, d);
        if (!rc && pairs && (he = st.d_pair.reserve(n + 1)) != hipSuccess) rc = fail(e, BSW_E_NOMEM, "device staging: %s", hipGetErrorString(he));
        if (!rc) rc = enqueue_batch(e, dp, p.variant, st.d_seq.p, st.d_tasks.p, st.d_order.p, ci.plan, st.d_out.p, stream, nullptr, fork_for(ctx, stream), pairs ? st.d_pair.p : nullptr);
        if (rc) return bail(rc);
        char *co = (char *)out + chunks[k].base * rec;
        pend.direct = is_registered(co, n * rec);
        if (!pend.direct && (he = st.h_out.reserve(n)) != hipSuccess) return bail(fail(e, BSW_E_NOMEM, "pinned staging: %s", hipGetErrorString(he)));
        /* A result copy queued behind its kernels sits at the head of its DMA engine's ring until they finish and holds up
         * the copies queued to that engine after it (profiles/r2/wire_submit_timeline.txt).  With the reference on the
         * device the input DMAs are short and that wait is what the pipeline loses (+7 % when the slot thread issues the
         * result DMA itself once the kernels are done); with 448 B per seed of input the link is busy anyway and the
         * extra host round trip per chunk costs more than it saves (-6 %), so there the copy is queued right away. */
        const bool late = rtasks != nullptr;
        if (!late) he = hipMemcpyAsync(pend.direct ? (void *)co : (void *)st.h_out.p, d_res(), n * rec, hipMemcpyDeviceToHost, stream);
        else he = hipEventRecord(dev.events[s], stream);
        if (he != hipSuccess) return bail(fail(e, BSW_E_HIP, "result DMA: %s", hipGetErrorString(he)));
        pend.copied = !late;
        pend.active = true; pend.n = n; pend.out = co;
        queued = false;                             /* from here on finish() / bail() drain through pend */
        if (dbg) { const double t4 = tnow(); t_staging += t1 - t0; t_host += t2 - t1; t_finish += t3 - t2; t_stage += t4 - t3; }
    }
    const double t5 = dbg ? tnow() : 0;
    const int rc = finish();
    if (dbg) fprintf(stderr, "[bsw] slot %zu.%zu: wait staging %.2f ms, host pass %.2f, wait results %.2f, DMA turn + enqueue %.2f, drain %.2f\n",
                     d, s, t_staging, t_host, t_finish, t_stage, tnow() - t5);
    return rc ? bail(rc) : BSW_OK;
}

static int submit_pipeline(bsw_ctx *ctx, bsw_params p, const bsw_task *tasks, const bsw_ref *ref, const bsw_ref_task *rtasks,
                           size_t n, bsw_result *out, bool packed = false)
{
    bsw_dparams dp;
    int rc = check_params(ctx->err, &p, &dp);
    if (rc) return rc;
    const size_t G = ctx->devs.size(), S = (size_t)ctx->cfg.streams;
    const std::vector<std::vector<chunk_span>> chunks = plan_chunks(n, ctx->cfg.chunk_tasks, G);
    struct wk { size_t d, s; int rc = 0; errs e; };
    std::vector<wk> ws;
    for (size_t s = 0; s < S; ++s)
        for (size_t d = 0; d < G; ++d)
            if (s < chunks[d].size()) { wk w; w.d = d; w.s = s; ws.push_back(w); }
    if (ws.empty()) return BSW_OK;
    const int gather_threads = std::max(1, ctx->cfg.pack_threads / (int)ws.size());
    std::atomic<int> abort_flag{0};
    std::vector<h2d_gate> gates(G);
    std::vector<std::thread> th;
    for (size_t k = 1; k < ws.size(); ++k)
        th.emplace_back([&, k]() { ws[k].rc = slot_worker(ctx, p, dp, tasks, ref, rtasks, out, chunks[ws[k].d], ws[k].d, ws[k].s, gather_threads, abort_flag, gates[ws[k].d], ws[k].e, packed); });
    ws[0].rc = slot_worker(ctx, p, dp, tasks, ref, rtasks, out, chunks[ws[0].d], ws[0].d, ws[0].s, gather_threads, abort_flag, gates[ws[0].d], ws[0].e, packed);
    for (auto &t : th) t.join();
    for (int pass = 0; pass < 2; ++pass)             /* report the failure itself, not the slots it made give up */
        for (auto &w : ws)
            if (w.rc && (pass || w.e.msg.compare(0, 7, "aborted") != 0)) { ctx->err = w.e; return w.rc; }
    return BSW_OK;
}

extern "C" int bsw_submit(bsw_ctx *ctx, const bsw_params *p, const bsw_task *tasks, size_t n, bsw_result *out)
{
    if (!ctx) return BSW_E_INVAL;
    if (!p || (!tasks && n) || (!out && n)) return fail(ctx->err, BSW_E_INVAL, "bsw_submit: NULL argument");
    if (ctx->dead) return fail(ctx->err, BSW_E_HIP, "bsw_submit: context is dead (an earlier wait for the GPU timed out)");
    if (ctx->worker_active) return fail(ctx->err, BSW_E_BUSY, "previous bsw_submit not waited for");
    bsw_dparams dp;
    int rc = check_params(ctx->err, p, &dp);
    if (rc) return rc;
    ctx->worker_active = true;
    ctx->worker_rc = 0;
    bsw_params pc = *p;
    ctx->worker = std::thread([ctx, pc, tasks, n, out]() { ctx->worker_rc = submit_pipeline(ctx, pc, tasks, nullptr, nullptr, n, out); });
    return BSW_OK;
}

/* bsw_submit for callers that keep their sequences 4-bit packed — 16 bases per uint64, base k in bits [4k, 4k+3], codes
 * 0-3 = ACGT, 4-7 = N, every sequence on an 8-byte boundary, lengths still in bases: the device's own layout and the
 * encoding the reference ships over its link (8 bases per 32-bit word, sw_pe_array_proc_element.v:1638,1677-1683).  The
 * words of a registered arena are DMA'd straight into the sequence buffer: no pack kernel, less than half the PCIe bytes
 * of byte-per-base input.  bsw_pack_bases() converts one sequence.  Wait with bsw_wait. */
extern "C" int bsw_submit_packed(bsw_ctx *ctx, const bsw_params *p, const bsw_task *tasks, size_t n, bsw_result *out)
{
    if (!ctx) return BSW_E_INVAL;
    if (!p || (!tasks && n) || (!out && n)) return fail(ctx->err, BSW_E_INVAL, "bsw_submit_packed: NULL argument");
    if (ctx->dead) return fail(ctx->err, BSW_E_HIP, "bsw_submit_packed: context is dead (an earlier wait for the GPU timed out)");
    if (ctx->worker_active) return fail(ctx->err, BSW_E_BUSY, "previous bsw_submit not waited for");
    bsw_dparams dp;
    int rc = check_params(ctx->err, p, &dp);
    if (rc) return rc;
    ctx->worker_active = true;
    ctx->worker_rc = 0;
    bsw_params pc = *p;
    ctx->worker = std::thread([ctx, pc, tasks, n, out]() { ctx->worker_rc = submit_pipeline(ctx, pc, tasks, nullptr, nullptr, n, out, true); });
    return BSW_OK;
}

/* bsw_submit for seeds against a DEVICE-RESIDENT reference (F3): only the reads cross PCIe; the targets are fetched
 * from the 2-bit pac on the GPU, the left flank of every read is mirrored by the pack kernel.  Wait with bsw_wait. */
extern "C" int bsw_submit_ref(bsw_ctx *ctx, const bsw_params *p, const bsw_ref *ref, const bsw_ref_task *rtasks, size_t n, bsw_result *out)
{
    if (!ctx) return BSW_E_INVAL;
    if (!p || !ref || (!rtasks && n) || (!out && n)) return fail(ctx->err, BSW_E_INVAL, "bsw_submit_ref: NULL argument");
    if (ref->d_pac.size() != ctx->devs.size()) return fail(ctx->err, BSW_E_INVAL, "bsw_submit_ref: the reference was uploaded through another context");
    if (ctx->dead) return fail(ctx->err, BSW_E_HIP, "bsw_submit_ref: context is dead (an earlier wait for the GPU timed out)");
    if (ctx->worker_active) return fail(ctx->err, BSW_E_BUSY, "previous bsw_submit not waited for");
    bsw_dparams dp;
    int rc = check_params(ctx->err, p, &dp);
    if (rc) return rc;
    ctx->worker_active = true;
    ctx->worker_rc = 0;
    bsw_params pc = *p;
    ctx->worker = std::thread([ctx, pc, ref, rtasks, n, out]() { ctx->worker_rc = submit_pipeline(ctx, pc, nullptr, ref, rtasks, n, out); });
    return BSW_OK;
}

extern "C" int bsw_wait(bsw_ctx *ctx)
{
    if (!ctx) return BSW_E_INVAL;
    if (!ctx->worker_active) return BSW_OK;
    if (ctx->worker.joinable()) ctx->worker.join();
    ctx->worker_active = false;
    return ctx->worker_rc;
}

/* ---- batched plain ksw_extend2 ------------------------------------------------ */
/* One pass each, per-task w / end_bonus / h0.  A task's band is min(w, max_ins, max_del) (sw_pe_array_sw_extend.v:
 * 1881,1890), which is exactly a per-task band limit, so tasks with different w and end_bonus share one launch:
 * P.w = the largest w of the group, wlim_r = min(w_i, gap limit of end_bonus_i). */
static int ext_group(bsw_ctx *ctx, errs &e, stage_t &st, hipStream_t s, hipEvent_t ev, const bsw_params *p,
                     const bsw_ext_task *tasks, const uint32_t *idx, size_t n, int w_group, bsw_ext *out)
{
    bsw_params pp = *p;
    pp.w = w_group;
    pp.max_band_try = 1;
    const int mx = mat_max(p->mat);
    std::vector<bsw_task> pt(n);
    for (size_t k = 0; k < n; ++k) {
        const bsw_ext_task &x = tasks[idx[k]];
        bsw_task &t = pt[k];
        memset(&t, 0, sizeof(t));
        if (x.qlen < 1) return fail(e, BSW_E_INVAL, "ext task %u: qlen must be >= 1", idx[k]);
        t.rquery = x.query; t.rtarget = x.target; t.rqlen = x.qlen; t.rtlen = x.tlen;
        t.h0 = x.h0; t.init_score = -1; t.tag = idx[k];
        const int gl = gap_limit(p, mx, x.qlen, x.end_bonus);
        t.wlim_r = x.w >= 1 ? std::min(x.w, gl) : 0;        /* w < 1 groups: P.w itself is the band */
    }
    bsw_dparams dp;
    int rc = check_params(e, &pp, &dp);
    if (rc) return rc;
    std::vector<bsw_result> res(n);
    rc = run_chunk(ctx, e, st, s, ev, pp, dp, pt.data(), n, res.data(), 1);
    if (rc) return rc;
    for (size_t k = 0; k < n; ++k) {
        out[idx[k]] = res[k].right;
        out[idx[k]].aw = tasks[idx[k]].w;
    }
    return BSW_OK;
}

static int ext_batch_on(bsw_ctx *ctx, errs &e, stage_t &st, hipStream_t s, hipEvent_t ev, const bsw_params *p,
                        const bsw_ext_task *tasks, size_t n, bsw_ext *out)
{
    std::vector<uint32_t> pos, odd;
    int wmax = 1;
    for (size_t i = 0; i < n; ++i) {
        if (tasks[i].w >= 1) { pos.push_back((uint32_t)i); wmax = std::max(wmax, tasks[i].w); }
        else odd.push_back((uint32_t)i);
    }
    if (wmax > (1 << 20)) return fail(e, BSW_E_INVAL, "band out of range");
    int rc = BSW_OK;
    const size_t chunk = std::max<size_t>(ctx->cfg.chunk_tasks, 1);
    for (size_t b0 = 0; b0 < pos.size() && !rc; b0 += chunk)
        rc = ext_group(ctx, e, st, s, ev, p, tasks, pos.data() + b0, std::min(chunk, pos.size() - b0), wmax, out);
    /* w <= 0 (never passed by bwa): one launch per distinct value */
    std::stable_sort(odd.begin(), odd.end(), [&](uint32_t a, uint32_t b) { return tasks[a].w < tasks[b].w; });
    for (size_t g0 = 0; g0 < odd.size() && !rc;) {
        size_t g1 = g0;
        while (g1 < odd.size() && tasks[odd[g1]].w == tasks[odd[g0]].w) ++g1;
        const int w = tasks[odd[g0]].w;
        if (w < 0) return fail(e, BSW_E_INVAL, "ext task %u: negative band", odd[g0]);
        rc = ext_group(ctx, e, st, s, ev, p, tasks, odd.data() + g0, g1 - g0, w, out);
        g0 = g1;
    }
    return rc;
}

extern "C" int bsw_extend_batch(bsw_ctx *ctx, const bsw_params *p, const bsw_ext_task *tasks, size_t n, bsw_ext *out)
{
    if (!ctx) return BSW_E_INVAL;
    errs &e = ctx->err;
    if (!p || (!tasks && n) || (!out && n)) return fail(e, BSW_E_INVAL, "bsw_extend_batch: NULL argument");
    int rc = busy_check(ctx, "bsw_extend_batch");
    if (rc) return rc;
    HIPCHK(e, hipSetDevice(ctx->device0()));
    return ext_batch_on(ctx, e, ctx->small, ctx->stream0(), ctx->devs[0].events[0], p, tasks, n, out);
}

/* ---- drop-in scalar ABI ----------------------------------------------------------
 * bwa calls ksw_extend2 from its -t worker threads.  Calls that arrive while a device round trip is in
 * flight are queued; the thread that finds no round trip in flight becomes the leader, takes everything
 * queued (its own call included), runs it as ONE device batch on the process-wide context's staging
 * (persistent pinned + device buffers: no allocation per call) and wakes the others. */
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
static std::mutex g_mu;
static std::condition_variable g_cv;
static std::vector<scalar_req *> g_queue;
static bool g_leader = false;
static bsw_ctx *g_ctx = nullptr;
static int g_ctx_rc = 0;
static std::atomic<int> g_variant{BSW_VARIANT_H};
static std::atomic<uint64_t> g_scalar_calls{0}, g_scalar_trips{0};

extern "C" void bsw_set_default_variant(int variant) { g_variant = variant == BSW_VARIANT_M ? BSW_VARIANT_M : BSW_VARIANT_H; }

/* calls served and device round trips made by the scalar ABI so far (calls / trips = mean coalescing factor) */
extern "C" void bsw_scalar_stats(uint64_t *calls, uint64_t *trips)
{
    if (calls) *calls = g_scalar_calls;
    if (trips) *trips = g_scalar_trips;
}

static bool same_scoring(const bsw_params &a, const bsw_params &b)
{
    return memcmp(a.mat, b.mat, 25) == 0 && a.o_del == b.o_del && a.e_del == b.e_del && a.o_ins == b.o_ins &&
           a.e_ins == b.e_ins && a.zdrop == b.zdrop && a.variant == b.variant;
}

static bool same_alignment_scoring(const bsw_params &a, const bsw_params &b)
{
    return memcmp(a.mat, b.mat, 25) == 0 && a.o_del == b.o_del && a.e_del == b.e_del && a.o_ins == b.o_ins && a.e_ins == b.e_ins;
}

/* A batch API rejects the WHOLE batch on its first bad task (a query beyond the class limits, an unknown xtra bit ...).
 * Calls of different threads share a batch here, so one thread's over-limit call must not fail the others: when a group of
 * several calls comes back with a per-task error (BSW_E_LIMIT / BSW_E_INVAL) its members are rerun one by one and only the
 * offender keeps the error. */
static bool per_task_error(int rc) { return rc == BSW_E_LIMIT || rc == BSW_E_INVAL; }

static int scalar_align_group(std::vector<scalar_req *> &batch, const std::vector<size_t> &grp, bool quiet)
{
    std::vector<bsw_atask> t(grp.size());
    std::vector<bsw_kswr> o(grp.size());
    for (size_t k = 0; k < grp.size(); ++k) t[k] = batch[grp[k]]->at;
    const int rc = bsw_align_batch(g_ctx, &batch[grp[0]]->p, t.data(), t.size(), o.data());
    if (rc && !quiet) fprintf(stderr, "ksw_align2(libbwasw_mi355): GPU path failed (%d): %s\n", rc, bsw_last_error(g_ctx));
    for (size_t k = 0; k < grp.size(); ++k) { batch[grp[k]]->rc = rc; if (!rc) batch[grp[k]]->ar = o[k]; }
    return rc;
}

static int scalar_global_group(std::vector<scalar_req *> &batch, const std::vector<size_t> &grp, bool quiet)
{
    int cap = 0;
    for (size_t k : grp) cap = std::max(cap, batch[k]->cap);
    std::vector<bsw_gtask> t(grp.size());
    std::vector<bsw_gresult> o(grp.size());
    std::vector<uint32_t> cg(cap ? grp.size() * (size_t)cap : 1);
    for (size_t k = 0; k < grp.size(); ++k) t[k] = batch[grp[k]]->gt;
    const int rc = bsw_global_batch(g_ctx, &batch[grp[0]]->p, t.data(), t.size(), cap, o.data(), cap ? cg.data() : nullptr);
    if (rc && !quiet) fprintf(stderr, "ksw_global2(libbwasw_mi355): GPU path failed (%d): %s\n", rc, bsw_last_error(g_ctx));
    for (size_t k = 0; k < grp.size(); ++k) {
        scalar_req *r = batch[grp[k]];
        r->rc = rc;
        if (rc) continue;
        r->gr = o[k];
        if (r->cap && o[k].n_cigar > 0) r->cg.assign(cg.begin() + (ptrdiff_t)(k * (size_t)cap), cg.begin() + (ptrdiff_t)(k * (size_t)cap) + o[k].n_cigar);
    }
    return rc;
}

static int scalar_extend_group(std::vector<scalar_req *> &batch, const std::vector<size_t> &grp, bool quiet)
{
    std::vector<bsw_ext_task> t(grp.size());
    std::vector<bsw_ext> x(grp.size());
    for (size_t k = 0; k < grp.size(); ++k) t[k] = batch[grp[k]]->t;
    int rc = BSW_OK;
    if (hipSetDevice(g_ctx->device0()) != hipSuccess) rc = BSW_E_HIP;
    if (!rc) rc = ext_batch_on(g_ctx, g_ctx->err, g_ctx->small, g_ctx->stream0(), g_ctx->devs[0].events[0], &batch[grp[0]]->p, t.data(), t.size(), x.data());
    if (rc && !quiet) fprintf(stderr, "ksw_extend2(libbwasw_mi355): GPU path failed (%d): %s\n", rc, bsw_last_error(g_ctx));
    for (size_t k = 0; k < grp.size(); ++k) { batch[grp[k]]->rc = rc; if (!rc) batch[grp[k]]->x = x[k]; }
    return rc;
}

/* one group of calls that share their scoring: one device batch; per-task errors are isolated to their callers */
static void scalar_run_group(std::vector<scalar_req *> &batch, const std::vector<size_t> &grp, int kind)
{
    const auto run = [&](const std::vector<size_t> &g, bool quiet) {
        return kind == 1 ? scalar_align_group(batch, g, quiet) : kind == 2 ? scalar_global_group(batch, g, quiet) : scalar_extend_group(batch, g, quiet);
    };
    const int rc = run(grp, grp.size() > 1);
    if (rc && grp.size() > 1) {
        if (!per_task_error(rc)) {                 /* a device failure: everybody's, reported once */
            fprintf(stderr, "%s(libbwasw_mi355): GPU path failed (%d): %s\n", kind == 1 ? "ksw_align2" : kind == 2 ? "ksw_global2" : "ksw_extend2", rc, bsw_last_error(g_ctx));
            return;
        }
        for (size_t k : grp) run(std::vector<size_t>{k}, false);
    }
}

/* the calls of one trip, grouped by kind and scoring: one bsw_extend / bsw_align_batch / bsw_global_batch launch sequence per
 * group (ksw_align2 / ksw_global2 used to be one serialised device round trip per call) */
static void scalar_round_trip(std::vector<scalar_req *> &batch)
{
    if (!g_ctx && !g_ctx_rc) {
        bsw_config c;
        bsw_default_config(&c);
        const char *dv = getenv("BSW_DEVICE");
        if (dv) c.device = atoi(dv);
        c.kernel = BSW_KERNEL_WAVE;              /* a handful of seeds per trip: one wavefront per extension */
        g_ctx_rc = bsw_create(&c, &g_ctx);
        if (g_ctx_rc) fprintf(stderr, "ksw_extend2(libbwasw_mi355): cannot create GPU context (%d); no CPU fallback exists\n", g_ctx_rc);
    }
    if (!g_ctx) {
        for (scalar_req *r : batch) r->rc = g_ctx_rc;
        return;
    }
    if (batch.empty()) return;                     /* (called only to create the context) */
    ++g_scalar_trips;
    g_scalar_calls += batch.size();
    std::vector<char> taken(batch.size(), 0);
    for (int pass = 0; pass < 2; ++pass)           /* the alignment kinds first, then the extensions (as before) */
        for (size_t i = 0; i < batch.size(); ++i) {
            const int kind = batch[i]->kind;
            if (taken[i] || (pass == 0) != (kind != 0)) continue;
            std::vector<size_t> grp;
            for (size_t j = i; j < batch.size(); ++j)
                if (!taken[j] && batch[j]->kind == kind &&
                    (kind == 0 ? same_scoring(batch[i]->p, batch[j]->p) : same_alignment_scoring(batch[i]->p, batch[j]->p))) { grp.push_back(j); taken[j] = 1; }
            scalar_run_group(batch, grp, kind);
        }
}

/* queue the call; whoever finds no trip in flight becomes the leader, takes everything queued and runs it */
static void scalar_call(scalar_req &req)
{
    std::unique_lock<std::mutex> lk(g_mu);
    g_queue.push_back(&req);
    while (!req.done) {
        if (!g_leader) {
            g_leader = true;
            std::vector<scalar_req *> batch;
            batch.swap(g_queue);
            lk.unlock();
            scalar_round_trip(batch);
            lk.lock();
            for (scalar_req *r : batch) r->done = true;
            g_leader = false;
            g_cv.notify_all();
        } else {
            g_cv.wait(lk);
        }
    }
}

extern "C" int ksw_extend2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat,
                           int o_del, int e_del, int o_ins, int e_ins, int w, int end_bonus, int zdrop, int h0,
                           int *qle, int *tle, int *gtle, int *gscore, int *max_off)
{
    auto neutral = [&](int score) {
        if (qle) *qle = 0;
        if (tle) *tle = 0;
        if (gtle) *gtle = 0;
        if (gscore) *gscore = -1;
        if (max_off) *max_off = 0;
        return score;
    };
    if (m != 5 || !mat || (qlen > 0 && !query) || (tlen > 0 && !target)) {
        fprintf(stderr, "ksw_extend2(libbwasw_mi355): unsupported arguments (m must be 5)\n");
        return neutral(-1);
    }
    if (h0 <= 0 || qlen <= 0) return neutral(h0 > 0 ? h0 : 0);      /* outside bwa's assert(h0 > 0) domain */
    if (tlen < 0) tlen = 0;
    scalar_req req;
    bsw_default_params(&req.p);
    memcpy(req.p.mat, mat, 25);
    req.p.o_del = o_del; req.p.e_del = e_del; req.p.o_ins = o_ins; req.p.e_ins = e_ins;
    req.p.zdrop = zdrop; req.p.variant = g_variant;
    memset(&req.t, 0, sizeof(req.t));
    req.t.query = query; req.t.target = target; req.t.qlen = qlen; req.t.tlen = tlen;
    req.t.w = w; req.t.end_bonus = end_bonus; req.t.h0 = h0;
    scalar_call(req);
    if (req.rc) return neutral(-1);
    if (qle) *qle = req.x.qle;
    if (tle) *tle = req.x.tle;
    if (gtle) *gtle = req.x.gtle;
    if (gscore) *gscore = req.x.gscore;
    if (max_off) *max_off = req.x.max_off;
    return req.x.score;
}

extern "C" int ksw_extend(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat,
                          int gapo, int gape, int w, int end_bonus, int zdrop, int h0,
                          int *qle, int *tle, int *gtle, int *gscore, int *max_off)
{
    return ksw_extend2(qlen, query, tlen, target, m, mat, gapo, gape, gapo, gape, w, end_bonus, zdrop, h0,
                       qle, tle, gtle, gscore, max_off);
}

/* ---- reference wire format end to end (F1) -----------------------------------------
 * bsw_refbatch_submit queues 256 KiB task batches; bsw_refbatch_wait parses the 8-word headers on the host
 * (task_parse.v:1931-1940), DMAs the batches as they are, unpacks the nibble streams on the GPU
 * (bsw_wire_pack_kernel) and runs everything queued as one device batch. */
extern "C" int bsw_refbatch_submit(bsw_ctx *ctx, const uint32_t *in_words, uint32_t *out_words)
{
    if (!ctx) return BSW_E_INVAL;
    if (!in_words || !out_words) return fail(ctx->err, BSW_E_INVAL, "bsw_refbatch_submit: NULL argument");
    int rc = busy_check(ctx, "bsw_refbatch_submit");
    if (rc) return rc;
    if (ctx->ref_queue.size() >= BSW_REFBATCH_MAX_INFLIGHT) return fail(ctx->err, BSW_E_BUSY, "%d task batches already in flight", BSW_REFBATCH_MAX_INFLIGHT);
    if (in_words[2] > BSW_REFBATCH_MAX_TASKS) return fail(ctx->err, BSW_E_LIMIT, "task batch announces %u tasks (> %d)", in_words[2], BSW_REFBATCH_MAX_TASKS);
    ctx->ref_queue.push_back(refbatch_req{in_words, out_words});
    return BSW_OK;
}

/* one run of queued batches [q0, q1) that share G0/G1 */
/* queued batches [q0, q1) (same scoring header) -> one device batch on stream s, nothing waited for: headers parsed on
 * host threads, batches DMA'd as they are, nibble streams unpacked on the GPU.  *n_out = tasks enqueued (0: nothing
 * in flight, the result batches are already written). */
/* wire batches per device batch and device batches in flight: measured best at 16 x 4 (profiles/r2/wire_format_rate.jsonl).
 * 16 batches = ~13 k seeds stay below the lane kernels' minimum batch on purpose: a lane launch costs one wave's full
 * duration (1.6 ms per side) however few seeds it holds, the wave-per-seed kernel finishes such a group sooner. */
#define REFBATCH_GROUP 16
#define REFBATCH_SLOTS 4
static int refbatch_enqueue(bsw_ctx *ctx, size_t q0, size_t q1, int variant, int zdrop, stage_t &st, hipStream_t s, size_t *n_out)
{
    errs &e = ctx->err;
    *n_out = 0;
    const uint32_t *W0 = ctx->ref_queue[q0].in;
    bsw_params p;
    bsw_default_params(&p);                       /* matrix a=1,b=4,N=-1 is hard-wired (sw_extend.v:1915-1940) */
    p.o_del = (int)(W0[0] & 0xff); p.e_del = (int)((W0[0] >> 8) & 0xff);
    p.o_ins = (int)((W0[0] >> 16) & 0xff); p.e_ins = (int)((W0[0] >> 24) & 0xff);
    p.pen_clip5 = (int)(W0[1] & 0xff); p.pen_clip3 = (int)((W0[1] >> 8) & 0xff);
    p.w = (int)((W0[1] >> 16) & 0xff);
    p.zdrop = zdrop; p.variant = variant; p.max_band_try = 2;
    bsw_dparams dp;
    int rc = check_params(e, &p, &dp);
    if (rc) return rc;
    size_t n = 0;
    for (size_t q = q0; q < q1; ++q) n += ctx->ref_queue[q].in[2];
    if (n == 0) {
        for (size_t q = q0; q < q1; ++q) memset(ctx->ref_queue[q].out, 0, BSW_REFBATCH_OUT_WORDS * sizeof(uint32_t));
        return BSW_OK;
    }
    const size_t nb = q1 - q0, wire_words = nb * (size_t)BSW_REFBATCH_IN_WORDS;
    hipError_t he;
    if ((he = st.h_tasks.reserve(n + 1)) != hipSuccess || (he = st.h_woff.reserve(n + 1)) != hipSuccess ||
        (he = st.h_out.reserve(n + 1)) != hipSuccess || (he = st.h_raw.reserve(wire_words * 4 + RAW_SLACK)) != hipSuccess)
        return fail(e, BSW_E_NOMEM, "pinned staging: %s", hipGetErrorString(he));
    /* headers -> task records (host: 8 words per task) */
    chunk_info ci;
    rc = fill_binparams(e, &p, ctx->cfg.kernel, ci.bp);
    if (rc) return rc;
    bsw_binparams &bp = ci.bp;
    uint32_t cw_all[BSW_MAX_WAVE_CLASSES] = {0}, cw[BSW_MAX_WAVE_CLASSES] = {0};
    uint32_t cl[BSW_MAX_LANE_CLASSES] = {0}, cr[BSW_MAX_LANE_CLASSES] = {0}, n_lane = 0;
    uint64_t acc = 0;
    /* pass 1 (cheap): where every batch's tasks and sequence words start */
    std::vector<uint64_t> wbase(nb + 1, 0), tbase(nb + 1, 0);
    for (size_t q = q0; q < q1; ++q) {
        const uint32_t *W = ctx->ref_queue[q].in;
        const uint32_t nt = W[2];
        uint64_t words = 0;
        for (uint32_t i = 0; i < nt; ++i) {
            const uint32_t *H = &W[8 + 8 * i];
            const int lq = (int)(H[0] & 0xff), lt = (int)((H[0] >> 16) & 0x7ff), rq = (int)(H[1] & 0xff), rt = (int)((H[1] >> 16) & 0x7ff);
            words += (lq ? nwords(lq) + nwords(lt) : 0) + (rq ? nwords(rq) + nwords(rt) : 0);
        }
        wbase[q - q0 + 1] = wbase[q - q0] + words;
        tbase[q - q0 + 1] = tbase[q - q0] + nt;
    }
    acc = wbase[nb];
    /* pass 2 (parallel over batches): records, class counts, and the batch itself into pinned staging */
    struct part { uint32_t cw_all[BSW_MAX_WAVE_CLASSES] = {0}, cw[BSW_MAX_WAVE_CLASSES] = {0}, cl[BSW_MAX_LANE_CLASSES] = {0}, cr[BSW_MAX_LANE_CLASSES] = {0}, n_lane = 0; int rc = 0; errs e; };
    const size_t nth = std::max<size_t>(1, std::min<size_t>((size_t)ctx->cfg.pack_threads, nb / 4));
    std::vector<part> parts(nth);
    auto parse = [&](size_t t) {
        part &pt = parts[t];
        for (size_t q = q0 + t; q < q1; q += nth) {
            const uint32_t *W = ctx->ref_queue[q].in;
            const uint32_t nt = W[2];
            uint64_t a2 = wbase[q - q0];
            size_t ti = (size_t)tbase[q - q0];
            const int64_t base = nt ? (int64_t)(8 + 8 * nt) - (int64_t)W[8 + 2] : 0;
            for (uint32_t i = 0; i < nt; ++i, ++ti) {
                const uint32_t *H = &W[8 + 8 * i];
                bsw_dtask &d = st.h_tasks.p[ti];
                bsw_wireoff &wo = st.h_woff.p[ti];
                memset(&d, 0, sizeof(d));
                const int lq = (int)(H[0] & 0xff), lt = (int)((H[0] >> 16) & 0x7ff), rq = (int)(H[1] & 0xff), rt = (int)((H[1] >> 16) & 0x7ff);
                const int64_t pos = base + (int64_t)H[2];
                if (pos < 8 + 8 * (int64_t)nt || pos + (lq + rq + lt + rt + 7) / 8 > BSW_REFBATCH_IN_WORDS) {
                    pt.rc = fail(pt.e, BSW_E_INVAL, "malformed task batch (task %u: data position)", i);
                    return;
                }
                const int h0 = (int)(H[4] & 0xff);
                if (h0 <= 0) { pt.rc = fail(pt.e, BSW_E_INVAL, "task batch: task %u has h0 <= 0", i); return; }
                wo.nib = ((uint64_t)(q - q0) * BSW_REFBATCH_IN_WORDS + (uint64_t)pos) * 8u;
                wo.lqlen = (uint16_t)lq; wo.rqlen = (uint16_t)rq; wo.ltlen = (uint16_t)lt; wo.rtlen = (uint16_t)rt;
                if (lq) { d.lq_off = (uint32_t)a2; a2 += nwords(lq); d.lt_off = (uint32_t)a2; a2 += nwords(lt); }
                if (rq) { d.rq_off = (uint32_t)a2; a2 += nwords(rq); d.rt_off = (uint32_t)a2; a2 += nwords(rt); }
                d.lqlen = (uint16_t)lq; d.rqlen = (uint16_t)rq; d.ltlen = (uint16_t)lt; d.rtlen = (uint16_t)rt;
                /* H5/H6 = {max_del[31:16], max_ins[15:0]}: the band limit the RTL applies (proc_element.v:925,933).  A
                 * non-positive limit (never written by bwa: both are >= 1) reads as 1, exactly as bsw_refbatch_decode maps it,
                 * so a malformed header gives the same band through either entry point */
                auto lim = [](uint32_t h) {
                    const int mi = (int)(int16_t)(h & 0xffff), md = (int)(int16_t)(h >> 16);
                    const int l = mi < md ? mi : md;
                    return (uint16_t)(l < 1 ? 1 : l);
                };
                d.wlim_l = lim(H[5]); d.wlim_r = lim(H[6]);
                d.h0 = h0; d.init_score = (int)(int16_t)(H[3] & 0xffff); d.qbeg = (int)(H[3] >> 16); d.tag = H[7];
                const int wc = bsw_wave_class_of(&bp, std::max(lq, rq));
                ++pt.cw_all[wc];
                const int bits = bsw_seed_lane_bits(&bp, lq, rq, h0);
                if (!bits) ++pt.cw[wc];
                else { ++pt.n_lane; if (lq) ++pt.cl[bsw_side_lane_class(&bp, bits, lq)]; if (rq) ++pt.cr[bsw_side_lane_class(&bp, bits, rq)]; }
            }
            memcpy((uint32_t *)st.h_raw.p + (q - q0) * (size_t)BSW_REFBATCH_IN_WORDS, W, BSW_REFBATCH_IN_WORDS * sizeof(uint32_t));
        }
    };
    {
        std::vector<std::thread> th;
        for (size_t t = 1; t < nth; ++t) th.emplace_back(parse, t);
        parse(0);
        for (auto &x : th) x.join();
    }
    for (const part &pt : parts) {
        if (pt.rc) { e = pt.e; return pt.rc; }
        for (int c = 0; c < BSW_MAX_WAVE_CLASSES; ++c) { cw_all[c] += pt.cw_all[c]; cw[c] += pt.cw[c]; }
        for (int c = 0; c < BSW_MAX_LANE_CLASSES; ++c) { cl[c] += pt.cl[c]; cr[c] += pt.cr[c]; }
        n_lane += pt.n_lane;
    }
    if (ctx->cfg.kernel == BSW_KERNEL_AUTO && !lane_bins_pay(n_lane, cl, cr)) bp.lane_on = 0;
    if (bp.lane_on && narrow_foldable(bp)) narrow_fold(bp, cl, cr, nullptr);     /* (wire-format groups: one launch per side) */
    if (!bp.lane_on) { memcpy(cw, cw_all, sizeof(cw)); memset(cl, 0, sizeof(cl)); memset(cr, 0, sizeof(cr)); n_lane = 0; }
    batch_plan &pl = ci.plan;
    pl = batch_plan();
    for (int c = 0; c < BSW_MAX_WAVE_CLASSES; ++c) pl.wave_start[c + 1] = pl.wave_start[c] + cw[c];
    uint32_t cur = pl.wave_start[BSW_MAX_WAVE_CLASSES];
    pl.lane_all_off = cur; pl.lane_all_cnt = n_lane; cur += n_lane;
    for (int c = 0; c < BSW_MAX_LANE_CLASSES; ++c) { pl.laneL_off[c] = cur; cur += cl[c]; }
    pl.laneL_off[BSW_MAX_LANE_CLASSES] = cur;
    for (int c = 0; c < BSW_MAX_LANE_CLASSES; ++c) { pl.laneR_off[c] = cur; cur += cr[c]; }
    pl.laneR_off[BSW_MAX_LANE_CLASSES] = cur;
    pl.redo_off = cur; pl.order_len = cur + n_lane;
    pl.redo_cls = bsw_wave_class_of(&bp, std::max(bp.cols8, bp.cols16) - 1);
    /* (pl.dep stays all ones: the wire-format groups run their classes on one stream) */
    memcpy(bp.wave_start, pl.wave_start, sizeof(bp.wave_start));
    bp.lane_all_off = pl.lane_all_off;
    memcpy(bp.laneL_off, pl.laneL_off, sizeof(bp.laneL_off));
    memcpy(bp.laneR_off, pl.laneR_off, sizeof(bp.laneR_off));
    /* device: wire batches -> seq, bins, DP kernels, results */
    if ((he = st.d_raw.reserve(wire_words * 4 + RAW_SLACK)) != hipSuccess || (he = st.d_seq.reserve((size_t)acc + 4)) != hipSuccess ||
        (he = st.d_tasks.reserve(n + 1)) != hipSuccess || (he = st.d_woff.reserve(n + 1)) != hipSuccess ||
        (he = st.d_order.reserve(order_capacity(n))) != hipSuccess || (he = st.d_bins.reserve(BSW_BIN_WORDS)) != hipSuccess ||
        (he = st.d_out.reserve(n + 1)) != hipSuccess)
        return fail(e, BSW_E_NOMEM, "device staging: %s", hipGetErrorString(he));
    HIPCHK(e, hipMemcpyAsync(st.d_raw.p, st.h_raw.p, wire_words * 4, hipMemcpyHostToDevice, s));
    HIPCHK(e, hipMemcpyAsync(st.d_tasks.p, st.h_tasks.p, n * sizeof(bsw_dtask), hipMemcpyHostToDevice, s));
    HIPCHK(e, hipMemcpyAsync(st.d_woff.p, st.h_woff.p, n * sizeof(bsw_wireoff), hipMemcpyHostToDevice, s));
    HIPCHK(e, bsw::launch_wire_pack((const uint32_t *)st.d_raw.p, st.d_tasks.p, st.d_woff.p, (uint32_t)n, st.d_seq.p, s));
    HIPCHK(e, bsw::launch_bin(bp, st.d_seq.p, st.d_tasks.p, (uint32_t)n, st.d_bins.p, st.d_order.p, s));
    rc = enqueue_batch(e, dp, variant, st.d_seq.p, st.d_tasks.p, st.d_order.p, pl, st.d_out.p, s, nullptr);
    if (rc) return rc;
    /* the result DMA is issued by refbatch_collect once the kernels are done: a copy queued now would sit in its DMA
     * engine's ring until then and hold up the next group's input copies queued behind it */
    *n_out = n;
    return BSW_OK;
}

/* wait for an enqueued group and write its 16 KiB result batches (host threads, one batch at a time each) */
static int refbatch_collect(bsw_ctx *ctx, size_t q0, size_t q1, size_t n, stage_t &st, hipStream_t s, hipEvent_t ev)
{
    errs &e = ctx->err;
    {
        const int rc0 = sync_stream(ctx, e, s, ev);
        if (rc0) return rc0;
        HIPCHK(e, hipMemcpyAsync(st.h_out.p, st.d_out.p, n * sizeof(bsw_result), hipMemcpyDeviceToHost, s));
    }
    static const bool dbg = getenv("BSW_DEBUG_TIMING") != nullptr;
    const auto tc0 = std::chrono::steady_clock::now();
    int rc = sync_stream(ctx, e, s, ev);
    if (rc) return rc;
    if (dbg) fprintf(stderr, "[bsw] wire:   waited %.3f ms for the GPU\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tc0).count());
    const size_t nb = q1 - q0;
    std::vector<size_t> tbase(nb + 1, 0);
    for (size_t q = q0; q < q1; ++q) tbase[q - q0 + 1] = tbase[q - q0] + ctx->ref_queue[q].in[2];
    const size_t nth = std::max<size_t>(1, std::min<size_t>((size_t)ctx->cfg.pack_threads, nb / 4));
    std::vector<int> trc(nth, 0);
    auto enc = [&](size_t t) {
        for (size_t q = q0 + t; q < q1; q += nth) {
            const uint32_t nt = ctx->ref_queue[q].in[2];
            memset(ctx->ref_queue[q].out, 0, BSW_REFBATCH_OUT_WORDS * sizeof(uint32_t));
            const int r = bsw_refbatch_encode_results(st.h_out.p + tbase[q - q0], nt, ctx->ref_queue[q].out);
            if (r < 0) trc[t] = r;
        }
    };
    std::vector<std::thread> th;
    for (size_t t = 1; t < nth; ++t) th.emplace_back(enc, t);
    enc(0);
    for (auto &x : th) x.join();
    for (int r : trc)
        if (r < 0) return fail(e, r, "result batch encode");
    return BSW_OK;
}

extern "C" int bsw_refbatch_wait(bsw_ctx *ctx, int variant, int zdrop)
{
    if (!ctx) return BSW_E_INVAL;
    int rc = busy_check(ctx, "bsw_refbatch_wait");
    if (rc) { ctx->ref_queue.clear(); return rc; }
    errs &e = ctx->err;
    if (variant != BSW_VARIANT_H && variant != BSW_VARIANT_M) { ctx->ref_queue.clear(); return fail(e, BSW_E_INVAL, "bad variant"); }
    if (hipSetDevice(ctx->device0()) != hipSuccess) { ctx->ref_queue.clear(); return fail(e, BSW_E_HIP, "hipSetDevice"); }
    const size_t nq = ctx->ref_queue.size();
    /* Runs of batches with the same scoring header become device batches of at most REFBATCH_GROUP wire batches, up to
     * REFBATCH_SLOTS of them in flight: the host parses the next group and writes an earlier group's result batches
     * while the others are on the GPU (the reference's manager keeps its four TBB/RBB pairs busy the same way,
     * batch_manager.v:418,745-773). */
    dev_state &dev = ctx->devs[0];
    static const size_t grp_tune = getenv("BSW_REFBATCH_GROUP") ? (size_t)std::max(1, atoi(getenv("BSW_REFBATCH_GROUP"))) : (size_t)REFBATCH_GROUP;   /* (measurements) */
    const size_t grp_env = grp_tune;
    const size_t NS = std::max<size_t>(1, std::min<size_t>(REFBATCH_SLOTS, dev.slots.size()));
    const size_t slot_of[REFBATCH_SLOTS] = {0, 1, 2, 3};
    struct flight { bool active = false; size_t q0 = 0, q1 = 0, n = 0; } fl[REFBATCH_SLOTS];
    static const bool dbg = getenv("BSW_DEBUG_TIMING") != nullptr;
    auto tnow = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_start = tnow();
    auto collect = [&](size_t sl) -> int {
        if (!fl[sl].active) return BSW_OK;
        fl[sl].active = false;
        const double t0 = tnow();
        const int r = refbatch_collect(ctx, fl[sl].q0, fl[sl].q1, fl[sl].n, dev.slots[slot_of[sl]], dev.streams[slot_of[sl]], dev.events[slot_of[sl]]);
        if (dbg) fprintf(stderr, "[bsw] wire: collect slot %zu batches [%zu,%zu): +%.3f .. +%.3f ms\n", sl, fl[sl].q0, fl[sl].q1, t0 - t_start, tnow() - t_start);
        return r;
    };
    auto drain = [&]() { for (size_t sl = 0; sl < NS; ++sl) if (fl[sl].active) { errs quiet; (void)sync_stream(ctx, quiet, dev.streams[slot_of[sl]], dev.events[slot_of[sl]]); fl[sl].active = false; } };
    size_t k = 0;
    for (size_t q0 = 0; q0 < nq;) {
        size_t q1 = q0 + 1;
        while (q1 < nq && q1 - q0 < grp_env && ctx->ref_queue[q1].in[0] == ctx->ref_queue[q0].in[0] && ctx->ref_queue[q1].in[1] == ctx->ref_queue[q0].in[1]) ++q1;
        if (nq - q1 < grp_env / 2)                                 /* no runt group at the end of a run */
            while (q1 < nq && ctx->ref_queue[q1].in[0] == ctx->ref_queue[q0].in[0] && ctx->ref_queue[q1].in[1] == ctx->ref_queue[q0].in[1]) ++q1;
        const size_t sl = k++ % NS;
        rc = collect(sl);
        size_t n_enq = 0;
        const double te = tnow();
        if (!rc) rc = refbatch_enqueue(ctx, q0, q1, variant, zdrop, dev.slots[slot_of[sl]], dev.streams[slot_of[sl]], &n_enq);
        if (dbg) fprintf(stderr, "[bsw] wire: enqueue slot %zu batches [%zu,%zu) %zu tasks: +%.3f .. +%.3f ms\n", sl, q0, q1, n_enq, te - t_start, tnow() - t_start);
        if (rc) {
            /* a failure half-way through refbatch_enqueue leaves copies out of the queued batches / kernels on this slot's
             * stream with fl[sl].active still false: drain that stream too before the caller gets its buffers back */
            { errs quiet; (void)sync_stream(ctx, quiet, dev.streams[slot_of[sl]], dev.events[slot_of[sl]]); }
            drain(); ctx->ref_queue.clear(); return rc;
        }
        if (n_enq) { fl[sl].active = true; fl[sl].q0 = q0; fl[sl].q1 = q1; fl[sl].n = n_enq; }
        q0 = q1;
    }
    for (size_t sl = 0; sl < NS; ++sl) {
        const size_t s2 = (k + sl) % NS;                                  /* oldest first */
        rc = collect(s2);
        if (rc) { drain(); ctx->ref_queue.clear(); return rc; }
    }
    ctx->ref_queue.clear();
    return (int)nq;
}

extern "C" int bsw_refbatch_run(bsw_ctx *ctx, const uint32_t *in_words, uint32_t *out_words, int variant, int zdrop)
{
    if (!ctx) return BSW_E_INVAL;
    if (!ctx->ref_queue.empty()) return fail(ctx->err, BSW_E_BUSY, "bsw_refbatch_run: task batches are queued (call bsw_refbatch_wait)");
    int rc = bsw_refbatch_submit(ctx, in_words, out_words);
    if (rc) return rc;
    const uint32_t n = in_words[2];
    rc = bsw_refbatch_wait(ctx, variant, zdrop);
    return rc < 0 ? rc : (int)n;
}

/* ---- banded global alignment with CIGAR (bwa ksw_global2) ------------------------------
 * Host side: lay the alignments out as right-side-only seeds so the byte-per-base sequences travel and are packed
 * exactly like extension tasks (registered arenas are DMA'd as they are), give every alignment its slice of the
 * backtrack matrix, sort by eh[] columns per lane, launch bsw_global_kernel, bring scores and CIGARs back. */
static int global_chunk(bsw_ctx *ctx, errs &e, const bsw_dparams &dp, const bsw_gtask *tasks, size_t n, int max_cigar,
                        bsw_gresult *res, uint32_t *cigars)
{
    stage_t &st = ctx->small;
    hipStream_t s = ctx->stream0();
    hipError_t he;
    if ((he = st.h_tasks.reserve(n + 1)) != hipSuccess || (he = st.h_roff.reserve(n + 1)) != hipSuccess)
        return fail(e, BSW_E_NOMEM, "pinned staging: %s", hipGetErrorString(he));
    std::vector<bsw_gdtask> gt(n);
    const int ncls = bsw::global_class_count();
    std::vector<uint32_t> order(n), cnt((size_t)ncls + 1, 0), cls(n);
    uint64_t acc = 0, accb = 0, zacc = 0;
    const uint8_t *lo = (const uint8_t *)UINTPTR_MAX, *hi = nullptr;
    for (size_t i = 0; i < n; ++i) {
        const bsw_gtask &t = tasks[i];
        bsw_dtask &d = st.h_tasks.p[i];
        bsw_rawoff &r = st.h_roff.p[i];
        memset(&d, 0, sizeof(d));
        memset(&r, 0, sizeof(r));
        d.rq_off = (uint32_t)acc; acc += nwords(t.qlen);
        d.rt_off = (uint32_t)acc; acc += nwords(t.tlen);
        d.rqlen = (uint16_t)t.qlen; d.rtlen = (uint16_t)t.tlen;
        r.rq = (uint32_t)accb; accb += (uint64_t)t.qlen;
        r.rt = (uint32_t)accb; accb += (uint64_t)t.tlen;
        if (t.qlen) { if (t.query < lo) lo = t.query; if (t.query + t.qlen > hi) hi = t.query + t.qlen; }
        if (t.tlen) { if (t.target < lo) lo = t.target; if (t.target + t.tlen > hi) hi = t.target + t.tlen; }
        bsw_gdtask &g = gt[i];
        g.q_off = d.rq_off; g.t_off = d.rt_off; g.qlen = t.qlen; g.tlen = t.tlen; g.w = t.w; g.pad = 0; g.z_off = zacc;
        const int n_col = t.qlen < 2 * t.w + 1 ? t.qlen : 2 * t.w + 1;
        if (cigars) zacc += (uint64_t)n_col * (uint64_t)t.tlen;
        int c = 0;
        while (c < ncls && t.qlen + 1 > bsw::global_class_cols(c)) ++c;
        cls[i] = (uint32_t)c;
        ++cnt[(size_t)c + 1];
    }
    for (int c = 0; c < ncls; ++c) cnt[(size_t)c + 1] += cnt[(size_t)c];
    {
        std::vector<uint32_t> pos(cnt.begin(), cnt.end() - 1);
        for (size_t i = 0; i < n; ++i) order[pos[cls[i]]++] = (uint32_t)i;
    }
    const size_t spanb = hi ? (size_t)(hi - lo) : 0;
    const bool direct = spanb > 0 && spanb < (1ull << 32) - RAW_SLACK && spanb <= 2 * accb + (1u << 20) && is_registered(lo, spanb);
    if (direct) {
        for (size_t i = 0; i < n; ++i) {
            bsw_rawoff &r = st.h_roff.p[i];
            r.rq = tasks[i].qlen ? (uint32_t)(tasks[i].query - lo) : 0;
            r.rt = tasks[i].tlen ? (uint32_t)(tasks[i].target - lo) : 0;
        }
    } else {
        if ((he = st.h_raw.reserve((size_t)accb + RAW_SLACK)) != hipSuccess) return fail(e, BSW_E_NOMEM, "pinned staging: %s", hipGetErrorString(he));
        for (size_t i = 0; i < n; ++i) {
            if (tasks[i].qlen) memcpy(st.h_raw.p + st.h_roff.p[i].rq, tasks[i].query, (size_t)tasks[i].qlen);
            if (tasks[i].tlen) memcpy(st.h_raw.p + st.h_roff.p[i].rt, tasks[i].target, (size_t)tasks[i].tlen);
        }
    }
    const size_t rawb = direct ? spanb : (size_t)accb;
    if ((he = st.d_raw.reserve(rawb + RAW_SLACK)) != hipSuccess || (he = st.d_seq.reserve((size_t)acc + 4)) != hipSuccess ||
        (he = st.d_tasks.reserve(n + 1)) != hipSuccess || (he = st.d_roff.reserve(n + 1)) != hipSuccess ||
        (he = ctx->g_tasks.reserve(n + 1)) != hipSuccess || (he = ctx->g_order.reserve(n + 1)) != hipSuccess ||
        (he = ctx->g_res.reserve(n + 1)) != hipSuccess || (cigars && (he = ctx->g_z.reserve((size_t)zacc + 64)) != hipSuccess) ||
        (cigars && (he = ctx->g_cig.reserve(n * (size_t)max_cigar + 1)) != hipSuccess))
        return fail(e, BSW_E_NOMEM, "device staging: %s", hipGetErrorString(he));
    if (rawb) HIPCHK(e, hipMemcpyAsync(st.d_raw.p, direct ? lo : st.h_raw.p, rawb, hipMemcpyHostToDevice, s));
    HIPCHK(e, hipMemcpyAsync(st.d_tasks.p, st.h_tasks.p, n * sizeof(bsw_dtask), hipMemcpyHostToDevice, s));
    HIPCHK(e, hipMemcpyAsync(st.d_roff.p, st.h_roff.p, n * sizeof(bsw_rawoff), hipMemcpyHostToDevice, s));
    HIPCHK(e, hipMemcpyAsync(ctx->g_tasks.p, gt.data(), n * sizeof(bsw_gdtask), hipMemcpyHostToDevice, s));
    HIPCHK(e, hipMemcpyAsync(ctx->g_order.p, order.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    HIPCHK(e, bsw::launch_pack(st.d_raw.p, st.d_tasks.p, st.d_roff.p, 0u, (uint32_t)n, 0, nullptr, 0, nullptr, st.d_seq.p, s));
    for (int c = 0; c < ncls; ++c) {
        const uint32_t k = cnt[(size_t)c + 1] - cnt[(size_t)c];
        if (!k) continue;
        HIPCHK(e, bsw::launch_global(c, dp, st.d_seq.p, ctx->g_tasks.p, ctx->g_order.p + cnt[(size_t)c], k,
                                     cigars ? ctx->g_z.p : nullptr, cigars ? ctx->g_cig.p : nullptr, max_cigar, ctx->g_res.p, s));
    }
    int rc = sync_stream(ctx, e, s, ctx->devs[0].events[0]);
    if (rc) return rc;
    HIPCHK(e, hipMemcpy(res, ctx->g_res.p, n * sizeof(bsw_gresult), hipMemcpyDeviceToHost));
    if (cigars) HIPCHK(e, hipMemcpy(cigars, ctx->g_cig.p, n * (size_t)max_cigar * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return BSW_OK;
}

extern "C" int bsw_global_batch(bsw_ctx *ctx, const bsw_params *p, const bsw_gtask *tasks, size_t n, int max_cigar,
                                bsw_gresult *res, uint32_t *cigars)
{
    if (!ctx) return BSW_E_INVAL;
    errs &e = ctx->err;
    if (!p || (!tasks && n) || (!res && n) || (cigars && max_cigar < 1)) return fail(e, BSW_E_INVAL, "bsw_global_batch: bad argument");
    int rc = busy_check(ctx, "bsw_global_batch");
    if (rc) return rc;
    bsw_params pp = *p;
    pp.w = 0;                                         /* the band is per task here */
    bsw_dparams dp;
    rc = check_params(e, &pp, &dp);
    if (rc) return rc;
    for (size_t i = 0; i < n; ++i) {
        const bsw_gtask &t = tasks[i];
        if (t.qlen < 0 || t.tlen < 0 || t.w < 0) return fail(e, BSW_E_INVAL, "global task %zu: negative length or band", i);
        if (t.qlen > BSW_MAX_QLEN || t.tlen > BSW_MAX_TLEN || t.w > BSW_MAX_TLEN) return fail(e, BSW_E_LIMIT, "global task %zu: beyond BSW_MAX_QLEN/BSW_MAX_TLEN", i);
        if ((t.qlen && !t.query) || (t.tlen && !t.target)) return fail(e, BSW_E_INVAL, "global task %zu: NULL sequence pointer", i);
    }
    HIPCHK(e, hipSetDevice(ctx->device0()));
    /* sub-batches: bounded backtrack memory (1 byte per banded cell) and sequence arena */
    const uint64_t zcap = 4ull << 30;
    for (size_t a = 0; a < n;) {
        size_t b = a;
        uint64_t zb = 0, sb = 0;
        while (b < n && b - a < (1u << 20)) {
            const bsw_gtask &t = tasks[b];
            const uint64_t nz = (uint64_t)(t.qlen < 2 * t.w + 1 ? t.qlen : 2 * t.w + 1) * (uint64_t)t.tlen;
            if (b > a && (zb + nz > zcap || sb + (uint64_t)(t.qlen + t.tlen) > (1ull << 31))) break;
            zb += cigars ? nz : 0;
            sb += (uint64_t)(t.qlen + t.tlen);
            ++b;
        }
        rc = global_chunk(ctx, e, dp, tasks + a, b - a, max_cigar, res + a, cigars ? cigars + a * (size_t)max_cigar : nullptr);
        if (rc) return rc;
        a = b;
    }
    return BSW_OK;
}

/* drop-in scalar ABI through the process-wide context: calls from concurrent threads share device round trips exactly
 * as ksw_extend2's do (one bsw_global_batch per trip and scoring).  Failure contract as ksw_extend2: message on stderr,
 * *n_cigar = 0, return -1. */
extern "C" int ksw_global2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat,
                           int o_del, int e_del, int o_ins, int e_ins, int w, int *n_cigar_, uint32_t **cigar_)
{
    if (n_cigar_) *n_cigar_ = 0;
    if (cigar_) *cigar_ = nullptr;
    if (m != 5 || !mat || qlen < 0 || tlen < 0 || (qlen > 0 && !query) || (tlen > 0 && !target)) {
        fprintf(stderr, "ksw_global2(libbwasw_mi355): unsupported arguments (m must be 5)\n");
        return -1;
    }
    scalar_req req;
    req.kind = 2;
    bsw_default_params(&req.p);
    memcpy(req.p.mat, mat, 25);
    req.p.o_del = o_del; req.p.e_del = e_del; req.p.o_ins = o_ins; req.p.e_ins = e_ins;
    memset(&req.gt, 0, sizeof(req.gt));
    req.gt.query = query; req.gt.target = target; req.gt.qlen = qlen; req.gt.tlen = tlen; req.gt.w = w < 0 ? 0 : w;
    const bool want = n_cigar_ && cigar_;
    req.cap = want ? qlen + tlen + 2 : 0;
    scalar_call(req);                                  /* coalesced with whatever other threads have queued */
    int score = -1;
    if (!req.rc) {
        score = req.gr.score;
        if (want && req.gr.n_cigar > 0) {
            *cigar_ = (uint32_t *)malloc((size_t)req.gr.n_cigar * sizeof(uint32_t));
            if (*cigar_) { memcpy(*cigar_, req.cg.data(), (size_t)req.gr.n_cigar * sizeof(uint32_t)); *n_cigar_ = req.gr.n_cigar; }
        }
    }
    return score;
}

extern "C" int ksw_global(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat,
                          int gapo, int gape, int w, int *n_cigar_, uint32_t **cigar_)
{
    return ksw_global2(qlen, query, tlen, target, m, mat, gapo, gape, gapo, gape, w, n_cigar_, cigar_);
}

/* ---- local alignment with start / second-best search (bwa ksw_align2, mate rescue) -----------
 * Host side as for the global alignment: the byte-per-base sequences travel and are packed like extension tasks
 * (registered arenas DMA'd as they are), every alignment gets its slice of the sub-optimal list scratch, tasks are
 * sorted by kernel class (mode x vectors per lane), bsw_align_kernel runs per class. */
static int align_chunk(bsw_ctx *ctx, errs &e, const bsw_dparams &dp, const bsw_atask *tasks, size_t n, bsw_kswr *out)
{
    stage_t &st = ctx->small;
    hipStream_t s = ctx->stream0();
    hipError_t he;
    if ((he = st.h_tasks.reserve(n + 1)) != hipSuccess || (he = st.h_roff.reserve(n + 1)) != hipSuccess)
        return fail(e, BSW_E_NOMEM, "pinned staging: %s", hipGetErrorString(he));
    std::vector<bsw_adtask> at(n);
    const int ncls = bsw::align_class_count();
    std::vector<uint32_t> order(n), cnt((size_t)ncls + 1, 0), cls(n);
    uint64_t acc = 0, accb = 0, bacc = 0;
    const uint8_t *lo = (const uint8_t *)UINTPTR_MAX, *hi = nullptr;
    for (size_t i = 0; i < n; ++i) {
        const bsw_atask &t = tasks[i];
        bsw_dtask &d = st.h_tasks.p[i];
        bsw_rawoff &r = st.h_roff.p[i];
        memset(&d, 0, sizeof(d));
        memset(&r, 0, sizeof(r));
        d.rq_off = (uint32_t)acc; acc += nwords(t.qlen);
        d.rt_off = (uint32_t)acc; acc += nwords(t.tlen);
        d.rqlen = (uint16_t)t.qlen; d.rtlen = (uint16_t)t.tlen;
        r.rq = (uint32_t)accb; accb += (uint64_t)t.qlen;
        r.rt = (uint32_t)accb; accb += (uint64_t)t.tlen;
        if (t.qlen) { if (t.query < lo) lo = t.query; if (t.query + t.qlen > hi) hi = t.query + t.qlen; }
        if (t.tlen) { if (t.target < lo) lo = t.target; if (t.target + t.tlen > hi) hi = t.target + t.tlen; }
        bsw_adtask &a = at[i];
        a.q_off = d.rq_off; a.t_off = d.rt_off; a.qlen = t.qlen; a.tlen = t.tlen; a.xtra = t.xtra; a.pad = 0; a.b_off = bacc;
        if (t.xtra & KSW_XSUBO) bacc += (uint64_t)t.tlen;
        const int c = bsw::align_class_of(t.qlen, (t.xtra & KSW_XBYTE) != 0);
        cls[i] = (uint32_t)c;
        ++cnt[(size_t)c + 1];
    }
    for (int c = 0; c < ncls; ++c) cnt[(size_t)c + 1] += cnt[(size_t)c];
    {
        std::vector<uint32_t> pos(cnt.begin(), cnt.end() - 1);
        for (size_t i = 0; i < n; ++i) order[pos[cls[i]]++] = (uint32_t)i;
    }
    const size_t spanb = hi ? (size_t)(hi - lo) : 0;
    const bool direct = spanb > 0 && spanb < (1ull << 32) - RAW_SLACK && spanb <= 2 * accb + (1u << 20) && is_registered(lo, spanb);
    if (direct) {
        for (size_t i = 0; i < n; ++i) {
            bsw_rawoff &r = st.h_roff.p[i];
            r.rq = tasks[i].qlen ? (uint32_t)(tasks[i].query - lo) : 0;
            r.rt = tasks[i].tlen ? (uint32_t)(tasks[i].target - lo) : 0;
        }
    } else {
        if ((he = st.h_raw.reserve((size_t)accb + RAW_SLACK)) != hipSuccess) return fail(e, BSW_E_NOMEM, "pinned staging: %s", hipGetErrorString(he));
        for (size_t i = 0; i < n; ++i) {
            if (tasks[i].qlen) memcpy(st.h_raw.p + st.h_roff.p[i].rq, tasks[i].query, (size_t)tasks[i].qlen);
            if (tasks[i].tlen) memcpy(st.h_raw.p + st.h_roff.p[i].rt, tasks[i].target, (size_t)tasks[i].tlen);
        }
    }
    const size_t rawb = direct ? spanb : (size_t)accb;
    if ((he = st.d_raw.reserve(rawb + RAW_FRONT + RAW_SLACK)) != hipSuccess || (he = st.d_seq.reserve((size_t)acc + 4)) != hipSuccess ||
        (he = st.d_tasks.reserve(n + 1)) != hipSuccess || (he = st.d_roff.reserve(n + 1)) != hipSuccess ||
        (he = ctx->a_tasks.reserve(n + 1)) != hipSuccess || (he = ctx->g_order.reserve(n + 1)) != hipSuccess ||
        (he = ctx->a_res.reserve(n + 1)) != hipSuccess || (he = ctx->a_bl.reserve((size_t)bacc + 64)) != hipSuccess)
        return fail(e, BSW_E_NOMEM, "device staging: %s", hipGetErrorString(he));
    if (rawb) HIPCHK(e, hipMemcpyAsync(st.d_raw.p + RAW_FRONT, direct ? lo : st.h_raw.p, rawb, hipMemcpyHostToDevice, s));
    HIPCHK(e, hipMemcpyAsync(st.d_tasks.p, st.h_tasks.p, n * sizeof(bsw_dtask), hipMemcpyHostToDevice, s));
    HIPCHK(e, hipMemcpyAsync(st.d_roff.p, st.h_roff.p, n * sizeof(bsw_rawoff), hipMemcpyHostToDevice, s));
    HIPCHK(e, hipMemcpyAsync(ctx->a_tasks.p, at.data(), n * sizeof(bsw_adtask), hipMemcpyHostToDevice, s));
    HIPCHK(e, hipMemcpyAsync(ctx->g_order.p, order.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    HIPCHK(e, bsw::launch_pack(st.d_raw.p + RAW_FRONT, st.d_tasks.p, st.d_roff.p, 0u, (uint32_t)n, 0, nullptr, 0, nullptr, st.d_seq.p, s));
    for (int c = 0; c < ncls; ++c) {
        const uint32_t k = cnt[(size_t)c + 1] - cnt[(size_t)c];
        if (!k) continue;
        HIPCHK(e, bsw::launch_align(c, dp, st.d_seq.p, ctx->a_tasks.p, ctx->g_order.p + cnt[(size_t)c], k, ctx->a_bl.p, ctx->a_res.p, s));
    }
    int rc = sync_stream(ctx, e, s, ctx->devs[0].events[0]);
    if (rc) return rc;
    HIPCHK(e, hipMemcpy(out, ctx->a_res.p, n * sizeof(bsw_kswr), hipMemcpyDeviceToHost));
    return BSW_OK;
}

extern "C" int bsw_align_batch(bsw_ctx *ctx, const bsw_params *p, const bsw_atask *tasks, size_t n, bsw_kswr *out)
{
    if (!ctx) return BSW_E_INVAL;
    errs &e = ctx->err;
    if (!p || (!tasks && n) || (!out && n)) return fail(e, BSW_E_INVAL, "bsw_align_batch: NULL argument");
    int rc = busy_check(ctx, "bsw_align_batch");
    if (rc) return rc;
    bsw_params pp = *p;
    pp.w = 0; pp.variant = BSW_VARIANT_H;
    bsw_dparams dp;
    rc = check_params(e, &pp, &dp);
    if (rc) return rc;
    int mxs = 0;
    for (int i = 0; i < 25; ++i) mxs = std::max(mxs, (int)p->mat[i]);
    if (mxs <= 0) return fail(e, BSW_E_INVAL, "bsw_align_batch: the scoring matrix has no positive score");
    for (size_t i = 0; i < n; ++i) {
        const bsw_atask &t = tasks[i];
        if (t.qlen < 0 || t.tlen < 0) return fail(e, BSW_E_INVAL, "align task %zu: negative length", i);
        if (t.qlen > BSW_ALIGN_MAX_QLEN || t.tlen > BSW_MAX_TLEN) return fail(e, BSW_E_LIMIT, "align task %zu: beyond BSW_ALIGN_MAX_QLEN/BSW_MAX_TLEN", i);
        if ((t.qlen && !t.query) || (t.tlen && !t.target)) return fail(e, BSW_E_INVAL, "align task %zu: NULL sequence pointer", i);
        if (t.xtra & ~(0xffff | KSW_XBYTE | KSW_XSTOP | KSW_XSUBO | KSW_XSTART)) return fail(e, BSW_E_INVAL, "align task %zu: unknown xtra flag", i);
    }
    HIPCHK(e, hipSetDevice(ctx->device0()));
    for (size_t a = 0; a < n;) {                      /* sub-batches: bounded sequence arena and sub-optimal list scratch */
        size_t b = a;
        uint64_t sb = 0, bb = 0;
        while (b < n && b - a < (1u << 20)) {
            const bsw_atask &t = tasks[b];
            if (b > a && (sb + (uint64_t)(t.qlen + t.tlen) > (1ull << 31) || bb + (uint64_t)t.tlen > (1ull << 28))) break;
            sb += (uint64_t)(t.qlen + t.tlen);
            bb += (t.xtra & KSW_XSUBO) ? (uint64_t)t.tlen : 0;
            ++b;
        }
        rc = align_chunk(ctx, e, dp, tasks + a, b - a, out + a);
        if (rc) return rc;
        a = b;
    }
    return BSW_OK;
}

static kswr_t align_scalar(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat,
                           int o_del, int e_del, int o_ins, int e_ins, int xtra)
{
    kswr_t r = {0, -1, -1, -1, -1, -1, -1};
    if (m != 5 || !mat || (qlen > 0 && !query) || (tlen > 0 && !target) || qlen < 0 || tlen < 0) {
        fprintf(stderr, "ksw_align2(libbwasw_mi355): unsupported arguments (m must be 5)\n");
        r.score = -1;
        return r;
    }
    scalar_req req;
    req.kind = 1;
    bsw_default_params(&req.p);
    memcpy(req.p.mat, mat, 25);
    req.p.o_del = o_del; req.p.e_del = e_del; req.p.o_ins = o_ins; req.p.e_ins = e_ins;
    memset(&req.at, 0, sizeof(req.at));
    req.at.query = query; req.at.target = target; req.at.qlen = qlen; req.at.tlen = tlen; req.at.xtra = xtra;
    scalar_call(req);                                  /* coalesced with whatever other threads have queued */
    r.score = -1;
    if (!req.rc) { r.score = req.ar.score; r.te = req.ar.te; r.qe = req.ar.qe; r.score2 = req.ar.score2; r.te2 = req.ar.te2; r.tb = req.ar.tb; r.qb = req.ar.qb; }
    return r;
}

extern "C" kswr_t ksw_align2(int qlen, uint8_t *query, int tlen, uint8_t *target, int m, const int8_t *mat,
                             int o_del, int e_del, int o_ins, int e_ins, int xtra, void **qry)
{
    (void)qry;
    return align_scalar(qlen, query, tlen, target, m, mat, o_del, e_del, o_ins, e_ins, xtra);
}

extern "C" kswr_t ksw_align(int qlen, uint8_t *query, int tlen, uint8_t *target, int m, const int8_t *mat,
                            int gapo, int gape, int xtra, void **qry)
{
    (void)qry;
    return align_scalar(qlen, query, tlen, target, m, mat, gapo, gape, gapo, gape, xtra);
}
