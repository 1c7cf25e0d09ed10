/*
 * bsw_ctx.hip — host side of libbwasw_mi355.so: context, registered host memory, parameters, the watchdog.
 *
 * The host side plays the role of the reference's batch_manager.v + tbb.v + rbb.v (CSR/DSM handshake,
 * 256 KiB task batches in, 16 KiB result batches out, round-robin over 4 PE arrays:
 * batch_manager.v:358-739) on top of the HIP runtime.  The host only validates lengths, counts
 * seeds per kernel class and starts DMAs; packing (byte-per-base -> 16 bases per uint64) and
 * binning (the (qlen, tlen, band) bins of BASELINE.json) run on the GPU (bsw_stage_kernel.hip),
 * chunk k of a submit goes to device k mod n_devices, and the kernels write results in task order.
 * There is no CPU compute path here: every DP cell is evaluated by the HIP kernels.
 * This file: bsw_create / bsw_destroy, bsw_host_*, defaults and validation, bsw_pack_bases / bsw_pack_tasks.
 */
#include <strings.h>
#include "bsw_internal.h"

/* ---- watchdog: never block in the runtime without a deadline (the RTL documents an
 * inactivity timeout, bwa_mem_sw.v:84-101, but a wedged PE array leaves its busy bit set forever) ---- */
BSW_LOCAL int wait_event(bsw_ctx *ctx, errs &e, hipEvent_t ev);
BSW_LOCAL int sync_stream(bsw_ctx *ctx, errs &e, hipStream_t st, hipEvent_t ev)
{
    if (ctx->dead) return fail(e, BSW_E_HIP, "context is dead (an earlier wait for the GPU timed out)");
    HIPCHK(e, hipEventRecord(ev, st));
    return wait_event(ctx, e, ev);
}

BSW_LOCAL int wait_event(bsw_ctx *ctx, errs &e, hipEvent_t ev)
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

BSW_LOCAL bool is_registered(const void *p, size_t len)
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
    c->pack_threads = 8;              /* the host pass of a chunk runs on 1 + pack_threads / slots threads (bsw_batch.hip) */
    c->chunk_tasks = 0;               /* 0 = sized per submit by the seeds' work (bsw_batch.hip: submit_common) */
    c->n_devices = 0;
    c->timeout_ms = 0;                /* 0 = the library default: BSW_TIMEOUT_MS if set, else 120 s (bsw_effective_timeout_ms) */
    c->result_format = BSW_RESULT_FULL;
    c->pin_threads = 1;
}

extern "C" int bsw_abi_version(void) { return BSW_ABI_VERSION; }

/* the watchdog a context created from `cfg` runs with: an explicit bsw_config.timeout_ms wins, 0 (what bsw_default_config
 * writes) means BSW_TIMEOUT_MS when that is a positive number, else 120 s.  Host only, no GPU needed. */
extern "C" int bsw_effective_timeout_ms(const bsw_config *cfg)
{
    if (cfg && cfg->timeout_ms > 0) return cfg->timeout_ms;
    const char *t = getenv("BSW_TIMEOUT_MS");
    return t && atoi(t) > 0 ? atoi(t) : 120000;
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
                if (f.mode == 1) for (auto a : f.aux) if (a) { (void)hipStreamSynchronize(a); (void)hipStreamDestroy(a); }      /* (mode 2 borrows the slot streams) */
                if (f.ev_fork) (void)hipEventDestroy(f.ev_fork);
                if (f.ev_fork_r) (void)hipEventDestroy(f.ev_fork_r);
                if (f.ev_nlist) (void)hipEventDestroy(f.ev_nlist);
                if (f.flag_mem) (void)hipFree(f.flag_mem);
                for (auto ev : f.ev_left) if (ev) (void)hipEventDestroy(ev);
                for (auto ev : f.ev_right) if (ev) (void)hipEventDestroy(ev);
                for (auto ev : f.ev_link) if (ev) (void)hipEventDestroy(ev);
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
        for (auto &pr : ctx->hist) { (void)hipEventDestroy(pr.e0); (void)hipEventDestroy(pr.mid); (void)hipEventDestroy(pr.e1); }
        ctx->small.release();
        ctx->g_tasks.release(); ctx->g_z.release(); ctx->g_cig.release(); ctx->g_order.release(); ctx->g_res.release();
        ctx->a_tasks.release(); ctx->a_bl.release(); ctx->a_res.release();
    }
    delete ctx;
}

/* "0-3,8,10-11" -> set; returns the number of CPUs, < 0 on a malformed list */
BSW_LOCAL int parse_cpulist(const char *text, cpu_set_t *out)
{
    CPU_ZERO(out);
    int n = 0;
    const char *p = text;
    while (*p) {
        while (*p == ',' || *p == ' ' || *p == '\n' || *p == '\t') ++p;
        if (!*p) break;
        char *end = nullptr;
        const long lo = strtol(p, &end, 10);
        if (end == p || lo < 0) return -1;
        long hi = lo;
        p = end;
        if (*p == '-') {
            hi = strtol(p + 1, &end, 10);
            if (end == p + 1 || hi < lo) return -1;
            p = end;
        }
        for (long c = lo; c <= hi; ++c)
            if (c < CPU_SETSIZE && !CPU_ISSET((int)c, out)) { CPU_SET((int)c, out); ++n; }
    }
    return n;
}

/* The CPUs next to a GPU: hipDeviceGetPCIBusId -> /sys/bus/pci/devices/<bdf>/local_cpulist (BSW_SYSFS_PCI replaces the
 * directory: tests), intersected with what the process may run on.  One manager next to its arrays
 * (batch_manager.v:745-773); SURVEY.md §8e names NUMA-local host threads as the condition for scaling over 8 GPUs. */
static void locate_device(dev_state &d, int pin)
{
    CPU_ZERO(&d.cpus);
    d.n_cpus = 0;
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf), d.device) != hipSuccess) { (void)hipGetLastError(); return; }
    for (char *c = bdf; *c; ++c) *c = (char)tolower(*c);
    d.bdf = bdf;
    const char *root = getenv("BSW_SYSFS_PCI");
    const std::string dir = std::string(root ? root : "/sys/bus/pci/devices") + "/" + bdf;
    char buf[4096];
    if (FILE *f = fopen((dir + "/numa_node").c_str(), "r")) {
        if (fgets(buf, sizeof(buf), f)) d.numa_node = atoi(buf);
        fclose(f);
    }
    if (pin < 0) return;
    cpu_set_t local, mine;
    int nl = 0;
    if (FILE *f = fopen((dir + "/local_cpulist").c_str(), "r")) {
        if (fgets(buf, sizeof(buf), f)) nl = parse_cpulist(buf, &local);
        fclose(f);
    }
    if (nl <= 0 || sched_getaffinity(0, sizeof(mine), &mine) != 0) return;
    CPU_AND(&d.cpus, &local, &mine);
    d.n_cpus = CPU_COUNT(&d.cpus);          /* 0: the process is confined to CPUs of another node — leave its threads alone */
}

BSW_LOCAL void pin_this_thread(const dev_state &d)
{
    if (d.n_cpus > 0) (void)pthread_setaffinity_np(pthread_self(), sizeof(cpu_set_t), &d.cpus);
}

extern "C" int bsw_device_placement(const bsw_ctx *ctx, int k, char *bdf, size_t bdf_cap, int *numa_node, int *n_cpus)
{
    if (!ctx || k < 0 || (size_t)k >= ctx->devs.size()) return BSW_E_INVAL;
    const dev_state &d = ctx->devs[(size_t)k];
    if (bdf && bdf_cap) { strncpy(bdf, d.bdf.c_str(), bdf_cap - 1); bdf[bdf_cap - 1] = 0; }
    if (numa_node) *numa_node = d.numa_node;
    if (n_cpus) *n_cpus = d.n_cpus;
    return BSW_OK;
}

extern "C" int bsw_create(const bsw_config *cfg, bsw_ctx **out) { return bsw_create_sized(cfg, sizeof(bsw_config), out); }

extern "C" int bsw_create_sized(const bsw_config *cfg, size_t cfg_size, bsw_ctx **out)
{
    if (!out) return BSW_E_INVAL;
    *out = nullptr;
    bsw_config c;
    bsw_default_config(&c);
    /* a caller built against an older header passes its own sizeof: the fields it does not know keep their defaults */
    if (cfg) {
        if (cfg_size < offsetof(bsw_config, timeout_ms) || cfg_size > sizeof(bsw_config)) return BSW_E_INVAL;
        memcpy(&c, cfg, cfg_size);
    }
    if (c.streams < 1) c.streams = 2;
    if (c.streams > 8) c.streams = 8;
    if (c.pack_threads < 1) c.pack_threads = 1;
    /* BSW_TIMEOUT_MS is the DEFAULT for hosts that pass no timeout of their own; an explicit bsw_config.timeout_ms wins */
    c.timeout_ms = bsw_effective_timeout_ms(&c);
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
        locate_device(d, c.pin_threads);
        d.slots.resize((size_t)c.streams);
        for (int s = 0; s < c.streams; ++s) {
            hipStream_t st = nullptr;
            hipEvent_t ev = nullptr;
            if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { ctx_release(ctx); return BSW_E_HIP; }
            d.streams.push_back(st);
            {
                /* The lane launches of a chunk on several streams.  Default (mode 2, stream 0 of a device only — the stream
                 * resident batches and synchronous chunks run on): a CHAIN, each launch released when every workgroup of the
                 * one before it has started (enqueue_parts; DESIGN.md §4.1b: 250 bp, the 232-column launches' ragged ends).
                 * BSW_FORK=1: every stream, all classes of a side released at once (round 4's first version:
                 * 1 971 GCUPS on 250 bp against 1 974 unforked, gpurun_out/r4h).  BSW_FORK=0: none. */
                static const int fork_env = getenv("BSW_FORK") ? atoi(getenv("BSW_FORK")) : -1;
                /* (the chain's waiting wave is bounded — bsw_wait_count gives up after 20 ms and the follower starts early, which
                 * is correct: the flag is a scheduling hint, data dependencies are events — so nothing here looks at profiler or
                 * launch-serialising environment variables any more; BSW_FORK=0 is the manual switch) */
                fork_t f;
                f.mode = fork_env == 1 ? 1 : (fork_env < 0 && s == 0 ? 2 : 0);
                bool good = f.mode != 0;
                /* the auxiliary streams run at the LOWEST priority: the widest class of a side (the slot stream's) has the
                 * longest waves and must get its slots first — released at the same instant, the narrow class's many short
                 * workgroups took half the slots and the long waves started late (right side 2.8 ms instead of 2.0,
                 * gpurun_out/r4c trace) */
                int least = 0, greatest = 0;
                static const bool noprio = getenv("BSW_FORK_NOPRIO") != nullptr;
                if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess || noprio) least = 0;
                /* (mode 2 BORROWS the device's other slot streams further down: they are idle whenever stream 0 runs a resident
                 * batch or a synchronous chunk, they sit on hardware queues of their own, and three more streams created
                 * here would push the slot streams onto shared queues — the runtime deals streams onto its four queues in
                 * creation order, and every pipeline leg of bench.py lost 5 - 20 % when that happened, gpurun_out/r7g) */
                for (int a = 0; a < BSW_FORK_AUX && good && f.mode == 1; ++a) good = hipStreamCreateWithPriority(&f.aux[a], hipStreamNonBlocking, least) == hipSuccess;
                good = good && hipEventCreateWithFlags(&f.ev_fork, hipEventDisableTiming) == hipSuccess &&
                       hipEventCreateWithFlags(&f.ev_fork_r, hipEventDisableTiming) == hipSuccess &&
                       hipEventCreateWithFlags(&f.ev_nlist, hipEventDisableTiming) == hipSuccess;
                for (int c = 0; c < BSW_MAX_LANE_CLASSES && good; ++c)
                    good = hipEventCreateWithFlags(&f.ev_left[c], hipEventDisableTiming) == hipSuccess &&
                           hipEventCreateWithFlags(&f.ev_right[c], hipEventDisableTiming) == hipSuccess;
                for (int k = 0; k < 2 * BSW_MAX_LANE_CLASSES && good && f.mode == 2; ++k)
                    good = hipEventCreateWithFlags(&f.ev_link[k], hipEventDisableTiming) == hipSuccess;
                /* (one more line behind the flags: how many waiting waves gave up at their deadline, bsw_chain_timeouts) */
                if (good && f.mode == 2) good = hipMalloc((void **)&f.flag_mem, (2 * BSW_MAX_LANE_CLASSES + 1) * 64 * sizeof(uint32_t)) == hipSuccess &&
                                                hipMemset(f.flag_mem, 0, (2 * BSW_MAX_LANE_CLASSES + 1) * 64 * sizeof(uint32_t)) == hipSuccess;
                if (!good) (void)hipGetLastError();     /* (optional machinery: its failure is not the next launch's error) */
                f.ok = good;
                d.forks.push_back(f);              /* (not ok: the classes of a side run one after the other on the slot stream) */
            }
            if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { ctx_release(ctx); return BSW_E_HIP; }
            d.events.push_back(ev);
            if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { ctx_release(ctx); return BSW_E_HIP; }
            d.h2d_done.push_back(ev);
        }
        if (!d.forks.empty() && d.forks[0].ok && d.forks[0].mode == 2) {
            fork_t &f = d.forks[0];
            f.naux = 0;
            for (size_t k = 1; k < d.streams.size() && f.naux < BSW_FORK_AUX; ++k) f.aux[f.naux++] = d.streams[k];
            if (f.naux == 0) f.ok = false;          /* (a context of one stream: nothing to chain on) */
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
    pipeline_shutdown(ctx);
    ctx_release(ctx);
}

/* ---- validation ---------------------------------------------------------------- */
BSW_LOCAL int check_params(errs &e, const bsw_params *p, bsw_dparams *dp)
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

BSW_LOCAL int mat_max(const int8_t *mat)
{
    int mx = 0;                                  /* bwa starts the scan at 0 */
    for (int i = 0; i < 25; ++i) mx = mx > mat[i] ? mx : mat[i];
    return mx;
}

/* min(max_ins, max_del): the longest useful gap (sw_pe_array_proc_element.v:925,933 H5/H6);
 * integer form of (int)((double)(qlen*max+end_bonus-o)/e + 1.) */
BSW_LOCAL int gap_limit(const bsw_params *p, int mx, int qlen, int end_bonus)
{
    int mi = (qlen * mx + end_bonus - p->o_ins + p->e_ins) / p->e_ins;
    int md = (qlen * mx + end_bonus - p->o_del + p->e_del) / p->e_del;
    if (mi < 1) mi = 1;
    if (md < 1) md = 1;
    int l = mi < md ? mi : md;
    return l > 65535 ? 65535 : l;
}

BSW_LOCAL size_t nwords(int len) { return (size_t)((len + 15) >> 4); }

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

BSW_LOCAL int busy_check(bsw_ctx *ctx, const char *what)
{
    if (ctx->dead) return fail(ctx->err, BSW_E_HIP, "%s: context is dead (an earlier wait for the GPU timed out)", what);
    if (pipeline_busy(ctx)) return fail(ctx->err, BSW_E_BUSY, "%s: a bsw_submit is in flight (call bsw_wait first)", what);
    return BSW_OK;
}

