/*
 * bsw_api.hip — host side of libbwasw_mi355.so: context, batch manager, C ABI.
 *
 * Plays the role of the reference's batch_manager.v + tbb.v + rbb.v (CSR/DSM handshake,
 * 256 KiB task batches in, 16 KiB result batches out, round-robin over 4 PE arrays:
 * batch_manager.v:358-739) on top of the HIP runtime: tasks are binned by the number of
 * eh[] columns a lane must hold, packed 16 bases per uint64 into pinned staging, streamed
 * with hipMemcpyAsync on several streams, and the kernels write results in task order.
 * There is no CPU compute path here: every DP cell is evaluated by the HIP kernels.
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "bsw_device.h"

namespace bsw {
int wave_class_count();
int wave_class_cols(int cls);
hipError_t launch_wave(int cls, int variant, const bsw_dparams &P, const uint64_t *seq, const bsw_dtask *tasks,
                       const uint32_t *order, uint32_t n, const uint32_t *n_dev, bsw_result *out, hipStream_t s);
int lane_class_count();
int lane_class_cols(int cls);
int lane_class_bits(int cls);
hipError_t launch_lane(int cls, int variant, const bsw_dparams &P, int side, const uint64_t *seq, const bsw_dtask *tasks,
                       const uint32_t *order, uint32_t n, bsw_result *out, hipStream_t s);
hipError_t launch_finalize(const bsw_dparams &P, const bsw_dtask *tasks, const uint32_t *order, uint32_t n,
                           bsw_result *out, uint32_t *redo, uint32_t *redo_cnt, hipStream_t s);
hipError_t launch_fetch(const uint8_t *pac, int64_t l_pac, const bsw_fetch_desc *desc, uint32_t nd, uint64_t *seq, hipStream_t s);
}  // namespace bsw

struct bsw_ref {
    uint8_t *d_pac = nullptr;
    int64_t l_pac = 0;
};

#define MAX_CLASSES 8
#define MAX_LANE_CLASSES 8
/* BSW_KERNEL_AUTO: a lane launch costs one wave's full duration (~1.3 ms for 150 bp seeds) however few seeds it
 * holds, the wave-per-task kernel scales with the seed count; measured crossover ~22k seeds (tools/crossover.py) */
#define LANE_AUTO_MIN 20000

/* How one batch is cut into kernel launches (all offsets index the device `order` array).
 *   [wave classes][lane seeds, any order][lane left sides by qlen][lane right sides by qlen][redo list] + counter */
struct batch_plan {
    uint32_t wave_start[MAX_CLASSES + 1] = {0};
    uint32_t lane_all_off = 0, lane_all_cnt = 0;
    uint32_t laneL_off[MAX_LANE_CLASSES + 1] = {0}, laneR_off[MAX_LANE_CLASSES + 1] = {0};
    uint32_t redo_off = 0;
    uint32_t order_len = 0;          /* entries before the redo counter */
    int redo_cls = 0;
};

struct slot_t;
static void slots_release(std::vector<slot_t> *v);

struct bsw_ctx {
    int device = 0;
    bsw_config cfg{};
    std::vector<hipStream_t> streams;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    bool timed = false;
    /* per-run event pairs since the last bsw_run_history() call (kernel time of every bsw_run) */
    std::vector<std::pair<hipEvent_t, hipEvent_t>> hist;
    size_t hist_used = 0;
    hipEvent_t ev_last0 = nullptr, ev_last1 = nullptr;
    std::string err;
    /* async submit */
    std::thread worker;
    bool worker_active = false;
    int worker_rc = 0;
    std::vector<struct slot_t> *slots = nullptr;    /* pinned + device staging, kept across submits */
};

struct bsw_dev_batch {
    uint64_t n = 0;
    bsw_dparams P{};
    int variant = 0;
    uint64_t *d_seq = nullptr;
    bsw_dtask *d_tasks = nullptr;
    uint32_t *d_order = nullptr;
    bsw_result *d_out = nullptr;
    uint64_t seq_words = 0;
    batch_plan plan;
    uint64_t launches = 0;
    uint64_t h2d_bytes = 0;      /* bytes the upload moved over PCIe */
};

static int fail(bsw_ctx *ctx, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
}

#define HIPCHK(ctx, call)                                                                         \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(ctx, BSW_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

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
    c->pack_threads = 16;
    c->chunk_tasks = 65536;
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

extern "C" const char *bsw_last_error(const bsw_ctx *ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

extern "C" int bsw_create(const bsw_config *cfg, bsw_ctx **out)
{
    if (!out) return BSW_E_INVAL;
    *out = nullptr;
    bsw_config c;
    if (cfg) c = *cfg; else bsw_default_config(&c);
    if (c.streams < 1) c.streams = 2;
    if (c.streams > 8) c.streams = 8;
    if (c.pack_threads < 1) c.pack_threads = 1;
    if (c.chunk_tasks == 0) c.chunk_tasks = 65536;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        fprintf(stderr, "libbwasw_mi355: no HIP device visible — this library has no CPU path\n");
        return BSW_E_NODEVICE;
    }
    if (c.device < 0 || c.device >= n) return BSW_E_INVAL;
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, c.device) != hipSuccess) return BSW_E_HIP;
    if (strncmp(pr.gcnArchName, "gfx950", 6) != 0) {
        fprintf(stderr, "libbwasw_mi355: device %d is %s, kernels are built for gfx950 only\n", c.device, pr.gcnArchName);
        return BSW_E_NODEVICE;
    }
    bsw_ctx *ctx = new bsw_ctx();
    ctx->device = c.device;
    ctx->cfg = c;
    if (hipSetDevice(c.device) != hipSuccess) { delete ctx; return BSW_E_HIP; }
    ctx->streams.resize((size_t)c.streams);
    for (auto &s : ctx->streams)
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { delete ctx; return BSW_E_HIP; }
    if (hipEventCreate(&ctx->ev_start) != hipSuccess || hipEventCreate(&ctx->ev_stop) != hipSuccess) { delete ctx; return BSW_E_HIP; }
    *out = ctx;
    return BSW_OK;
}

extern "C" void bsw_destroy(bsw_ctx *ctx)
{
    if (!ctx) return;
    if (ctx->worker_active && ctx->worker.joinable()) ctx->worker.join();
    (void)hipSetDevice(ctx->device);
    for (auto s : ctx->streams) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
    if (ctx->ev_start) (void)hipEventDestroy(ctx->ev_start);
    if (ctx->ev_stop) (void)hipEventDestroy(ctx->ev_stop);
    for (auto &pr : ctx->hist) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    slots_release(ctx->slots);
    delete ctx;
}

/* ---- validation + packing -------------------------------------------------- */
static int check_params(bsw_ctx *ctx, const bsw_params *p, bsw_dparams *dp)
{
    if (!p) return fail(ctx, BSW_E_INVAL, "params is NULL");
    if (p->e_del < 1 || p->e_ins < 1 || p->o_del < 0 || p->o_ins < 0)
        return fail(ctx, BSW_E_INVAL, "need e_del,e_ins >= 1 and o_del,o_ins >= 0");
    if (p->w < 0 || p->w > (1 << 20) || p->max_band_try > 8) return fail(ctx, BSW_E_INVAL, "band out of range");
    if (p->variant != BSW_VARIANT_H && p->variant != BSW_VARIANT_M) return fail(ctx, BSW_E_INVAL, "bad variant");
    if (p->o_del + p->e_del > 4096 || p->o_ins + p->e_ins > 4096) return fail(ctx, BSW_E_LIMIT, "gap penalties too large");
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

/* 8 base bytes (codes 0..4) -> 8 nibbles in the low 32 bits */
static inline uint64_t squeeze8(uint64_t x)
{
    x = (x | (x >> 4)) & 0x00FF00FF00FF00FFull;
    x = (x | (x >> 8)) & 0x0000FFFF0000FFFFull;
    x = (x | (x >> 16)) & 0x00000000FFFFFFFFull;
    return x;
}

/* byte-per-base -> 16 bases per uint64; returns non-zero when the sequence holds an N (code >= 4) */
static unsigned pack_seq(const uint8_t *s, int len, uint64_t *dst)
{
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

extern "C" int bsw_pack_bases(const uint8_t *bases, int len, uint64_t *words)
{
    if (len < 0 || (len > 0 && (!bases || !words))) return BSW_E_INVAL;
    return (int)pack_seq(bases, len, words);
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

struct packed_host {
    std::vector<uint64_t> seq;
    std::vector<bsw_dtask> tasks;
    std::vector<uint32_t> order;
    batch_plan plan;
};

static size_t order_capacity(size_t n) { return 4 * n + 16; }   /* upper bound of plan.order_len + 1 */

static int task_class(int qmax)
{
    const int nc = bsw::wave_class_count();
    for (int c = 0; c < nc; ++c)
        if (qmax + 1 <= bsw::wave_class_cols(c)) return c;
    return -1;
}

/* validate, lay out and pack tasks[0..n) — `threads` host threads do the nibble packing —
 * then bin them: the batch manager's (qlen, tlen, w) bins of BASELINE.json.
 *   - lane bins: seeds the lane-per-task kernel can take (bwa-style matrix, no N, short query,
 *     16-bit score range), each side sorted by query length so a wave holds equal-length queries;
 *   - wave classes: everything else, by the number of eh[] columns a lane must hold. */
static int pack_tasks(bsw_ctx *ctx, const bsw_params *p, const bsw_task *tasks, size_t n, int threads,
                      uint64_t *seq_dst /* may be NULL: use ph.seq */, size_t seq_cap, packed_host &ph,
                      bsw_dtask *task_dst, uint32_t *order_dst, size_t *seq_words_out,
                      bool dev_targets = false /* targets are fetched on the device: leave their words alone */)
{
    const int mx = mat_max(p->mat);
    static const bool dbg = getenv("BSW_DEBUG_TIMING") != nullptr;
    auto tnow = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_a = dbg ? tnow() : 0;
    std::vector<uint64_t> off(n + 1);
    uint64_t acc = 0;
    for (size_t i = 0; i < n; ++i) {
        const bsw_task &t = tasks[i];
        if (t.lqlen < 0 || t.rqlen < 0 || t.ltlen < 0 || t.rtlen < 0)
            return fail(ctx, BSW_E_INVAL, "task %zu: negative length", i);
        if (t.lqlen > BSW_MAX_QLEN || t.rqlen > BSW_MAX_QLEN || t.ltlen > BSW_MAX_TLEN || t.rtlen > BSW_MAX_TLEN)
            return fail(ctx, BSW_E_LIMIT, "task %zu: length beyond BSW_MAX_QLEN/BSW_MAX_TLEN", i);
        if (t.h0 <= 0) return fail(ctx, BSW_E_INVAL, "task %zu: h0 must be > 0", i);
        if ((int64_t)t.h0 + (int64_t)(t.lqlen + t.rqlen) * mx >= BSW_MAX_SCORE)
            return fail(ctx, BSW_E_LIMIT, "task %zu: score range beyond BSW_MAX_SCORE", i);
        if ((t.lqlen && (!t.lquery || (t.ltlen && !t.ltarget && !dev_targets))) ||
            (t.rqlen && (!t.rquery || (t.rtlen && !t.rtarget && !dev_targets))))
            return fail(ctx, BSW_E_INVAL, "task %zu: NULL sequence pointer", i);
        off[i] = acc;
        acc += (t.lqlen ? nwords(t.lqlen) + nwords(t.ltlen) : 0) + (t.rqlen ? nwords(t.rqlen) + nwords(t.rtlen) : 0);
    }
    off[n] = acc;
    if (acc >= (1ull << 32)) return fail(ctx, BSW_E_LIMIT, "batch sequence arena beyond 2^32 words; split the batch");
    uint64_t *seq = seq_dst;
    if (!seq) { ph.seq.assign((size_t)acc + 1, 0); seq = ph.seq.data(); }
    else if (acc > seq_cap) return fail(ctx, BSW_E_NOMEM, "staging too small");
    bsw_dtask *dt = task_dst;
    if (!dt) { ph.tasks.resize(n); dt = ph.tasks.data(); }
    uint32_t *ord = order_dst;
    if (!ord) { ph.order.assign(order_capacity(n), 0); ord = ph.order.data(); }

    double t_b = dbg ? tnow() : 0;
    auto work = [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
            const bsw_task &t = tasks[i];
            bsw_dtask &d = dt[i];
            uint64_t o = off[i];
            unsigned nn = 0;
            memset(&d, 0, sizeof(d));
            if (t.lqlen) {
                d.lq_off = (uint32_t)o; nn |= pack_seq(t.lquery, t.lqlen, seq + o); o += nwords(t.lqlen);
                d.lt_off = (uint32_t)o; if (!dev_targets) nn |= pack_seq(t.ltarget, t.ltlen, seq + o); o += nwords(t.ltlen);
            }
            if (t.rqlen) {
                d.rq_off = (uint32_t)o; nn |= pack_seq(t.rquery, t.rqlen, seq + o); o += nwords(t.rqlen);
                d.rt_off = (uint32_t)o; if (!dev_targets) nn |= pack_seq(t.rtarget, t.rtlen, seq + o); o += nwords(t.rtlen);
            }
            (void)nn;                               /* N-bearing seeds are fine for both kernels */
            d.lqlen = (uint16_t)t.lqlen; d.rqlen = (uint16_t)t.rqlen;
            d.ltlen = (uint16_t)t.ltlen; d.rtlen = (uint16_t)t.rtlen;
            d.wlim_l = (uint16_t)gap_limit(p, mx, t.lqlen, p->pen_clip5);
            d.wlim_r = (uint16_t)gap_limit(p, mx, t.rqlen, p->pen_clip3);
            d.h0 = t.h0; d.init_score = t.init_score; d.qbeg = t.qbeg; d.tag = t.tag;
        }
    };
    if (threads <= 1 || n < 4096) work(0, n);
    else {
        std::vector<std::thread> th;
        const size_t per = (n + (size_t)threads - 1) / (size_t)threads;
        for (int k = 0; k < threads; ++k) {
            const size_t lo = per * (size_t)k, hi = std::min(n, lo + per);
            if (lo < hi) th.emplace_back(work, lo, hi);
        }
        for (auto &t : th) t.join();
    }

    double t_c = dbg ? tnow() : 0;
    /* ---- binning ---- */
    batch_plan &pl = ph.plan;
    pl = batch_plan();
    const int nlc = bsw::lane_class_count();
    int cols8 = 0, cols16 = 0;                       /* widest lane class per value width */
    for (int c = 0; c < nlc; ++c) {
        if (bsw::lane_class_bits(c) == 8) cols8 = std::max(cols8, bsw::lane_class_cols(c));
        else cols16 = std::max(cols16, bsw::lane_class_cols(c));
    }
    const int kern = ctx ? ctx->cfg.kernel : BSW_KERNEL_AUTO;
    const bool lane_params = kern != BSW_KERNEL_WAVE && lane_matrix_ok(p);
    const int a = p->mat[0];
    std::vector<uint8_t> lane_bits(n ? n : 1, 0);    /* 0 = wave kernel, 8 / 16 = lane kernel value width */
    uint32_t n_lane = 0;
    if (lane_params) {
        for (size_t i = 0; i < n; ++i) {
            const bsw_task &t = tasks[i];
            const int qm = t.lqlen > t.rqlen ? t.lqlen : t.rqlen;
            const int64_t top = (int64_t)t.h0 + (int64_t)(t.lqlen + t.rqlen) * a;     /* no H can exceed this */
            if (top <= 255 && qm + 1 <= cols8) lane_bits[i] = 8;
            else if (top < 65000 && qm + 1 <= cols16) lane_bits[i] = 16;
            if (lane_bits[i]) ++n_lane;
        }
        if (kern == BSW_KERNEL_AUTO && n_lane < LANE_AUTO_MIN) {
            std::fill(lane_bits.begin(), lane_bits.end(), 0);
            n_lane = 0;
        }
    }
    auto side_class = [&](int bits, int q) {
        for (int c = 0; c < nlc; ++c)
            if (bsw::lane_class_bits(c) == bits && q + 1 <= bsw::lane_class_cols(c)) return c;
        return nlc - 1;
    };
    /* wave classes: counting sort by columns-per-lane class */
    uint32_t count[MAX_CLASSES] = {0};
    std::vector<uint8_t> cls(n ? n : 1, 0);
    for (size_t i = 0; i < n; ++i) {
        if (lane_bits[i]) continue;
        const bsw_task &t = tasks[i];
        const int c = task_class(t.lqlen > t.rqlen ? t.lqlen : t.rqlen);
        if (c < 0) return fail(ctx, BSW_E_LIMIT, "task %zu: no kernel class", i);
        cls[i] = (uint8_t)c;
        ++count[c];
    }
    uint32_t pos[MAX_CLASSES];
    pl.wave_start[0] = 0;
    for (int c = 0; c < MAX_CLASSES; ++c) pl.wave_start[c + 1] = pl.wave_start[c] + count[c];
    for (int c = 0; c < MAX_CLASSES; ++c) pos[c] = pl.wave_start[c];
    for (size_t i = 0; i < n; ++i)
        if (!lane_bits[i]) ord[pos[cls[i]]++] = (uint32_t)i;
    uint32_t cur = pl.wave_start[MAX_CLASSES];
    /* lane seeds (finalize pass) */
    pl.lane_all_off = cur;
    pl.lane_all_cnt = n_lane;
    for (size_t i = 0; i < n; ++i)
        if (lane_bits[i]) ord[cur++] = (uint32_t)i;
    /* per side: counting sort by (lane class, query length descending) so a wave holds equal-length queries */
    for (int side = 0; side < 2; ++side) {
        std::vector<uint32_t> hist((size_t)nlc * 256 + 1, 0);
        auto key = [&](uint32_t ti) -> int {
            const bsw_task &t = tasks[ti];
            const int q = side ? t.rqlen : t.lqlen;
            return q > 0 ? side_class(lane_bits[ti], q) * 256 + q : -1;
        };
        for (uint32_t k = 0; k < n_lane; ++k) {
            const int kk = key(ord[pl.lane_all_off + k]);
            if (kk >= 0) ++hist[(size_t)kk];
        }
        uint32_t *offs = side ? pl.laneR_off : pl.laneL_off;
        std::vector<uint32_t> start((size_t)nlc * 256 + 1, 0);
        uint32_t run = cur;
        for (int c = 0; c < nlc; ++c) {                 /* inside a class: longest queries (most work) first */
            offs[c] = run;
            for (int q = 255; q >= 0; --q) {
                start[(size_t)c * 256 + q] = run;
                run += hist[(size_t)c * 256 + q];
            }
        }
        for (int c = nlc; c <= MAX_LANE_CLASSES; ++c) offs[c] = run;
        for (uint32_t k = 0; k < n_lane; ++k) {
            const uint32_t ti = ord[pl.lane_all_off + k];
            const int kk = key(ti);
            if (kk >= 0) ord[start[(size_t)kk]++] = ti;
        }
        cur = run;
    }
    pl.redo_off = cur;
    pl.order_len = cur + n_lane;
    pl.redo_cls = task_class(std::max(cols8, cols16) - 1);
    if (dbg) fprintf(stderr, "[bsw] pack_tasks n=%zu: validate+alloc %.2f ms, pack %.2f ms, bin %.2f ms\n", n, t_b - t_a, t_c - t_b, tnow() - t_c);
    if (seq_words_out) *seq_words_out = (size_t)acc;
    return BSW_OK;
}

extern "C" int64_t bsw_plan_batch(const bsw_params *p, const bsw_task *tasks, size_t n, int kernel, int pack_threads,
                                  uint32_t *order, uint32_t *seg)
{
    if (!p || (!tasks && n) || !order || !seg) return BSW_E_INVAL;
    bsw_ctx tmp;                                   /* carries only the kernel choice; no device is touched */
    tmp.cfg.kernel = kernel;
    bsw_dparams dp;
    int rc = check_params(&tmp, p, &dp);
    if (rc) return rc;
    packed_host ph;
    size_t words = 0;
    rc = pack_tasks(&tmp, p, tasks, n, pack_threads > 0 ? pack_threads : 1, nullptr, 0, ph, nullptr, order, &words);
    if (rc) return rc;
    const batch_plan &pl = ph.plan;
    int k = 0;
    for (int c = 0; c < MAX_CLASSES; ++c) seg[k++] = pl.wave_start[c];
    seg[k++] = pl.lane_all_off;                    /* 8 */
    for (int c = 0; c < MAX_LANE_CLASSES; ++c) seg[k++] = pl.laneL_off[c];    /* 9..16 */
    for (int c = 0; c < MAX_LANE_CLASSES; ++c) seg[k++] = pl.laneR_off[c];    /* 17..24 */
    seg[k++] = pl.redo_off;                        /* 25 */
    seg[k++] = pl.order_len;                       /* 26 */
    return (int64_t)words;
}

/* ---- device-resident batches ------------------------------------------------ */
extern "C" void bsw_free_batch(bsw_ctx *ctx, bsw_dev_batch *b)
{
    if (!b) return;
    if (ctx) (void)hipSetDevice(ctx->device);
    if (b->d_seq) (void)hipFree(b->d_seq);
    if (b->d_tasks) (void)hipFree(b->d_tasks);
    if (b->d_order) (void)hipFree(b->d_order);
    if (b->d_out) (void)hipFree(b->d_out);
    delete b;
}

static int upload_common(bsw_ctx *ctx, const bsw_params *p, const bsw_task *tasks, size_t n, bsw_dev_batch **out,
                         const bsw_ref *ref /* NULL: targets come from the host */, const bsw_ref_task *rtasks)
{
    *out = nullptr;
    if (n >= (1ull << 32)) return fail(ctx, BSW_E_LIMIT, "more than 2^32-1 tasks in one batch");
    bsw_dparams dp;
    int rc = check_params(ctx, p, &dp);
    if (rc) return rc;
    packed_host ph;
    size_t words = 0;
    rc = pack_tasks(ctx, p, tasks, n, ctx->cfg.pack_threads, nullptr, 0, ph, nullptr, nullptr, &words, ref != nullptr);
    if (rc) return rc;
    std::vector<bsw_fetch_desc> descs;
    if (ref) {
        descs.reserve(2 * n);
        for (size_t i = 0; i < n; ++i) {
            const bsw_dtask &d = ph.tasks[i];
            const bsw_seed &sd = rtasks[i].seed;
            if (d.lqlen && d.ltlen) descs.push_back(bsw_fetch_desc{sd.rbeg - 1, d.lt_off, d.ltlen, -1, 0});
            if (d.rqlen && d.rtlen) descs.push_back(bsw_fetch_desc{sd.rbeg + sd.len, d.rt_off, d.rtlen, 1, 0});
        }
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    bsw_dev_batch *b = new bsw_dev_batch();
    b->n = n; b->P = dp; b->variant = p->variant; b->seq_words = words;
    b->plan = ph.plan;
    const size_t olen = (size_t)ph.plan.order_len + 1;          /* + redo counter */
    hipError_t e = hipSuccess;
    bsw_fetch_desc *d_desc = nullptr;
    if ((e = hipMalloc((void **)&b->d_seq, (words + 1) * sizeof(uint64_t))) != hipSuccess ||
        (e = hipMalloc((void **)&b->d_tasks, (n + 1) * sizeof(bsw_dtask))) != hipSuccess ||
        (e = hipMalloc((void **)&b->d_order, (olen + 1) * sizeof(uint32_t))) != hipSuccess ||
        (e = hipMalloc((void **)&b->d_out, (n + 1) * sizeof(bsw_result))) != hipSuccess ||
        (!descs.empty() && (e = hipMalloc((void **)&d_desc, descs.size() * sizeof(bsw_fetch_desc))) != hipSuccess)) {
        bsw_free_batch(ctx, b);
        return fail(ctx, BSW_E_NOMEM, "hipMalloc: %s", hipGetErrorString(e));
    }
    hipStream_t s = ctx->streams[0];
    if ((e = hipMemcpyAsync(b->d_seq, ph.seq.data(), words * sizeof(uint64_t), hipMemcpyHostToDevice, s)) != hipSuccess ||
        (e = hipMemcpyAsync(b->d_tasks, ph.tasks.data(), n * sizeof(bsw_dtask), hipMemcpyHostToDevice, s)) != hipSuccess ||
        (e = hipMemcpyAsync(b->d_order, ph.order.data(), olen * sizeof(uint32_t), hipMemcpyHostToDevice, s)) != hipSuccess ||
        (e = hipMemsetAsync(b->d_out, 0xff, n * sizeof(bsw_result), s)) != hipSuccess ||
        (!descs.empty() && ((e = hipMemcpyAsync(d_desc, descs.data(), descs.size() * sizeof(bsw_fetch_desc), hipMemcpyHostToDevice, s)) != hipSuccess ||
                            (e = bsw::launch_fetch(ref->d_pac, ref->l_pac, d_desc, (uint32_t)descs.size(), b->d_seq, s)) != hipSuccess)) ||
        (e = hipStreamSynchronize(s)) != hipSuccess) {
        if (d_desc) (void)hipFree(d_desc);
        bsw_free_batch(ctx, b);
        return fail(ctx, BSW_E_HIP, "upload: %s", hipGetErrorString(e));
    }
    if (d_desc) (void)hipFree(d_desc);
    b->h2d_bytes = words * 8 + n * sizeof(bsw_dtask) + olen * 4 + descs.size() * sizeof(bsw_fetch_desc);
    *out = b;
    return BSW_OK;
}

extern "C" int bsw_upload(bsw_ctx *ctx, const bsw_params *p, const bsw_task *tasks, size_t n, bsw_dev_batch **out)
{
    if (!ctx || !out || (!tasks && n)) return fail(ctx, BSW_E_INVAL, "bsw_upload: NULL argument");
    return upload_common(ctx, p, tasks, n, out, nullptr, nullptr);
}

/* ---- device-resident reference (F3) ------------------------------------------------ */
extern "C" int bsw_ref_upload(bsw_ctx *ctx, const uint8_t *pac, int64_t l_pac, bsw_ref **out)
{
    if (!ctx || !pac || !out || l_pac <= 0) return fail(ctx, BSW_E_INVAL, "bsw_ref_upload: bad argument");
    *out = nullptr;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    bsw_ref *r = new bsw_ref();
    r->l_pac = l_pac;
    const size_t bytes = (size_t)((l_pac + 3) >> 2);
    hipError_t e;
    if ((e = hipMalloc((void **)&r->d_pac, bytes + 8)) != hipSuccess) { delete r; return fail(ctx, BSW_E_NOMEM, "hipMalloc: %s", hipGetErrorString(e)); }
    if ((e = hipMemcpy(r->d_pac, pac, bytes, hipMemcpyHostToDevice)) != hipSuccess) {
        (void)hipFree(r->d_pac); delete r;
        return fail(ctx, BSW_E_HIP, "pac upload: %s", hipGetErrorString(e));
    }
    *out = r;
    return BSW_OK;
}

extern "C" void bsw_ref_free(bsw_ctx *ctx, bsw_ref *ref)
{
    if (!ref) return;
    if (ctx) (void)hipSetDevice(ctx->device);
    if (ref->d_pac) (void)hipFree(ref->d_pac);
    delete ref;
}

extern "C" int bsw_upload_ref(bsw_ctx *ctx, const bsw_params *p, const bsw_ref *ref, const bsw_ref_task *rt, size_t n, bsw_dev_batch **out)
{
    if (!ctx || !p || !ref || !out || (!rt && n)) return fail(ctx, BSW_E_INVAL, "bsw_upload_ref: NULL argument");
    /* mem_chain2aln's task extraction minus the target bases, which stay on the device */
    std::vector<bsw_task> tasks(n ? n : 1);
    size_t scratch_len = 0;
    for (size_t i = 0; i < n; ++i) scratch_len += (size_t)(rt[i].seed.qbeg > 0 ? rt[i].seed.qbeg : 0);
    std::vector<uint8_t> scratch(scratch_len + 1);
    size_t so = 0;
    const int64_t two = ref->l_pac << 1;
    for (size_t i = 0; i < n; ++i) {
        const bsw_ref_task &r = rt[i];
        const bsw_seed &sd = r.seed;
        if (!r.query || r.l_query < 1 || sd.qbeg < 0 || sd.len < 1 || sd.qbeg + sd.len > r.l_query)
            return fail(ctx, BSW_E_INVAL, "ref task %zu: bad seed / read", i);
        if (r.rmax0 < 0 || r.rmax1 > two || r.rmax0 > sd.rbeg || r.rmax1 < sd.rbeg + sd.len ||
            (r.rmax0 < ref->l_pac && ref->l_pac < r.rmax1))
            return fail(ctx, BSW_E_INVAL, "ref task %zu: window outside the reference or bridging the strands", i);
        const int64_t lt = sd.rbeg - r.rmax0, rtl = r.rmax1 - (sd.rbeg + sd.len);
        if (lt > BSW_MAX_TLEN || rtl > BSW_MAX_TLEN) return fail(ctx, BSW_E_LIMIT, "ref task %zu: window beyond BSW_MAX_TLEN", i);
        bsw_task &t = tasks[i];
        memset(&t, 0, sizeof(t));
        if (sd.qbeg > 0) {
            for (int k = 0; k < sd.qbeg; ++k) scratch[so + (size_t)k] = r.query[sd.qbeg - 1 - k];
            t.lquery = scratch.data() + so; t.lqlen = sd.qbeg; t.ltlen = (int32_t)lt;
            so += (size_t)sd.qbeg;
        }
        if (sd.qbeg + sd.len != r.l_query) {
            t.rquery = r.query + sd.qbeg + sd.len; t.rqlen = r.l_query - (sd.qbeg + sd.len); t.rtlen = (int32_t)rtl;
        }
        t.h0 = sd.len * p->mat[0]; t.init_score = r.init_score; t.qbeg = sd.qbeg; t.tag = r.tag;
    }
    return upload_common(ctx, p, tasks.data(), n, out, ref, rt);
}

extern "C" int bsw_extend_ref(bsw_ctx *ctx, const bsw_params *p, const bsw_ref *ref, const bsw_ref_task *rt, size_t n, bsw_result *out)
{
    if (!ctx || (!out && n)) return fail(ctx, BSW_E_INVAL, "bsw_extend_ref: NULL argument");
    bsw_dev_batch *b = nullptr;
    int rc = bsw_upload_ref(ctx, p, ref, rt, n, &b);
    if (rc) return rc;
    rc = bsw_run(ctx, b);
    if (!rc) rc = bsw_download(ctx, b, out);
    bsw_free_batch(ctx, b);
    return rc;
}

static int enqueue_batch(bsw_ctx *ctx, const bsw_dparams &P, int variant, const uint64_t *d_seq, const bsw_dtask *d_tasks,
                         uint32_t *d_order, const batch_plan &pl, bsw_result *d_out, hipStream_t s, uint64_t *launches)
{
    const int nc = bsw::wave_class_count();
    for (int c = 0; c < nc; ++c) {
        const uint32_t cnt = pl.wave_start[c + 1] - pl.wave_start[c];
        if (!cnt) continue;
        HIPCHK(ctx, bsw::launch_wave(c, variant, P, d_seq, d_tasks, d_order + pl.wave_start[c], cnt, nullptr, d_out, s));
        if (launches) ++*launches;
    }
    if (pl.lane_all_cnt) {
        uint32_t *redo_cnt = d_order + pl.order_len;
        HIPCHK(ctx, hipMemsetAsync(redo_cnt, 0, sizeof(uint32_t), s));
        const int nlc = bsw::lane_class_count();
        for (int side = 0; side < 2; ++side) {
            const uint32_t *offs = side ? pl.laneR_off : pl.laneL_off;
            for (int c = 0; c < nlc; ++c) {
                const uint32_t cnt = offs[c + 1] - offs[c];
                if (!cnt) continue;
                HIPCHK(ctx, bsw::launch_lane(c, variant, P, side, d_seq, d_tasks, d_order + offs[c], cnt, d_out, s));
                if (launches) ++*launches;
            }
        }
        HIPCHK(ctx, bsw::launch_finalize(P, d_tasks, d_order + pl.lane_all_off, pl.lane_all_cnt, d_out,
                                         d_order + pl.redo_off, redo_cnt, s));
        /* seeds whose first band try was not final: recompute from scratch, one wavefront each */
        HIPCHK(ctx, bsw::launch_wave(pl.redo_cls, variant, P, d_seq, d_tasks, d_order + pl.redo_off, pl.lane_all_cnt,
                                     redo_cnt, d_out, s));
        if (launches) *launches += 2;
    }
    return BSW_OK;
}

extern "C" int bsw_run(bsw_ctx *ctx, bsw_dev_batch *b)
{
    if (!ctx || !b) return fail(ctx, BSW_E_INVAL, "bsw_run: NULL argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->streams[0];
    hipEvent_t e0 = ctx->ev_start, e1 = ctx->ev_stop;
    if (ctx->hist_used < 4096) {
        if (ctx->hist_used == ctx->hist.size()) {
            hipEvent_t a, c;
            HIPCHK(ctx, hipEventCreate(&a));
            HIPCHK(ctx, hipEventCreate(&c));
            ctx->hist.emplace_back(a, c);
        }
        e0 = ctx->hist[ctx->hist_used].first;
        e1 = ctx->hist[ctx->hist_used].second;
        ++ctx->hist_used;
    }
    HIPCHK(ctx, hipEventRecord(e0, s));
    b->launches = 0;
    int rc = enqueue_batch(ctx, b->P, b->variant, b->d_seq, b->d_tasks, b->d_order, b->plan, b->d_out, s, &b->launches);
    if (rc) return rc;
    HIPCHK(ctx, hipEventRecord(e1, s));
    ctx->ev_last0 = e0; ctx->ev_last1 = e1;
    ctx->timed = true;
    return BSW_OK;
}

extern "C" int bsw_sync(bsw_ctx *ctx)
{
    if (!ctx) return BSW_E_INVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    for (auto s : ctx->streams) HIPCHK(ctx, hipStreamSynchronize(s));
    return BSW_OK;
}

extern "C" int bsw_last_run_ms(bsw_ctx *ctx, float *ms)
{
    if (!ctx || !ms || !ctx->timed) return BSW_E_INVAL;
    HIPCHK(ctx, hipEventSynchronize(ctx->ev_last1));
    HIPCHK(ctx, hipEventElapsedTime(ms, ctx->ev_last0, ctx->ev_last1));
    return BSW_OK;
}

extern "C" int bsw_run_history(bsw_ctx *ctx, float *ms, int cap)
{
    if (!ctx || (!ms && cap > 0)) return BSW_E_INVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int n = 0;
    for (size_t i = 0; i < ctx->hist_used && n < cap; ++i, ++n) {
        HIPCHK(ctx, hipEventSynchronize(ctx->hist[i].second));
        HIPCHK(ctx, hipEventElapsedTime(&ms[n], ctx->hist[i].first, ctx->hist[i].second));
    }
    ctx->hist_used = 0;
    return n;
}

extern "C" int bsw_download(bsw_ctx *ctx, bsw_dev_batch *b, bsw_result *out)
{
    if (!ctx || !b || (!out && b->n)) return fail(ctx, BSW_E_INVAL, "bsw_download: NULL argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->streams[0]));
    HIPCHK(ctx, hipMemcpy(out, b->d_out, b->n * sizeof(bsw_result), hipMemcpyDeviceToHost));
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

/* ---- streaming submit: pinned double buffering over several streams ----------- */
struct slot_t {
    uint64_t *h_seq = nullptr, *d_seq = nullptr; size_t seq_cap = 0;
    bsw_dtask *h_tasks = nullptr, *d_tasks = nullptr;
    uint32_t *h_order = nullptr, *d_order = nullptr;
    bsw_result *h_out = nullptr, *d_out = nullptr;
    size_t task_cap = 0;
    size_t base = 0, cnt = 0;       /* chunk in flight */
    bool busy = false;
};

static void slot_free(slot_t &s)
{
    if (s.h_seq) (void)hipHostFree(s.h_seq);
    if (s.d_seq) (void)hipFree(s.d_seq);
    if (s.h_tasks) (void)hipHostFree(s.h_tasks);
    if (s.d_tasks) (void)hipFree(s.d_tasks);
    if (s.h_order) (void)hipHostFree(s.h_order);
    if (s.d_order) (void)hipFree(s.d_order);
    if (s.h_out) (void)hipHostFree(s.h_out);
    if (s.d_out) (void)hipFree(s.d_out);
    s = slot_t();
}

static void slots_release(std::vector<slot_t> *v)
{
    if (!v) return;
    for (auto &s : *v) slot_free(s);
    delete v;
}

/* One staging slot = one stream = one host thread: chunk ci is handled by slot ci % nslots (validate, pack into
 * pinned staging, H2D, kernels, D2H, copy-out), so host packing of several chunks and the GPU work of several
 * chunks overlap — the round-robin of the reference's four TBB/RBB pairs (batch_manager.v:418,745-773). */
static int slot_worker(bsw_ctx *ctx, const bsw_params &p, const bsw_dparams &dp, const bsw_task *tasks, size_t n,
                       bsw_result *out, size_t k, size_t nslots, int threads, std::atomic<int> &abort_flag, std::string &err)
{
    bsw_ctx local;                       /* error text + kernel choice for this thread (ctx->err is not thread-safe) */
    local.cfg = ctx->cfg;
    auto failed = [&](int rc) { err = local.err; abort_flag = 1; return rc; };
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) { local.err = hipGetErrorString(e); return failed(BSW_E_HIP); }
    const size_t chunk = ctx->cfg.chunk_tasks;
    slot_t &s = (*ctx->slots)[k];
    hipStream_t st = ctx->streams[k];
    for (size_t base = k * chunk; base < n && !abort_flag; base += nslots * chunk) {
        const size_t cnt = std::min(chunk, n - base);
        size_t words = 0;
        for (size_t i = base; i < base + cnt; ++i) {
            const bsw_task &t = tasks[i];
            if (t.lqlen < 0 || t.rqlen < 0 || t.ltlen < 0 || t.rtlen < 0) { fail(&local, BSW_E_INVAL, "task %zu: negative length", i); return failed(BSW_E_INVAL); }
            words += (t.lqlen ? nwords(t.lqlen) + nwords(t.ltlen) : 0) + (t.rqlen ? nwords(t.rqlen) + nwords(t.rtlen) : 0);
        }
        if (s.seq_cap < words + 1) {
            if (s.h_seq) (void)hipHostFree(s.h_seq);
            if (s.d_seq) (void)hipFree(s.d_seq);
            s.h_seq = nullptr; s.d_seq = nullptr;
            s.seq_cap = (words + 1) * 5 / 4;
            if ((e = hipHostMalloc((void **)&s.h_seq, s.seq_cap * 8, hipHostMallocDefault)) != hipSuccess ||
                (e = hipMalloc((void **)&s.d_seq, s.seq_cap * 8)) != hipSuccess) { s.seq_cap = 0; fail(&local, BSW_E_NOMEM, "staging: %s", hipGetErrorString(e)); return failed(BSW_E_NOMEM); }
        }
        if (s.task_cap < cnt) {
            if (s.h_tasks) { (void)hipHostFree(s.h_tasks); (void)hipFree(s.d_tasks); (void)hipHostFree(s.h_order); (void)hipFree(s.d_order); (void)hipHostFree(s.h_out); (void)hipFree(s.d_out); }
            s.h_tasks = nullptr; s.d_tasks = nullptr; s.h_order = nullptr; s.d_order = nullptr; s.h_out = nullptr; s.d_out = nullptr;
            s.task_cap = std::max(cnt, chunk);
            if ((e = hipHostMalloc((void **)&s.h_tasks, s.task_cap * sizeof(bsw_dtask), hipHostMallocDefault)) != hipSuccess ||
                (e = hipMalloc((void **)&s.d_tasks, s.task_cap * sizeof(bsw_dtask))) != hipSuccess ||
                (e = hipHostMalloc((void **)&s.h_order, order_capacity(s.task_cap) * sizeof(uint32_t), hipHostMallocDefault)) != hipSuccess ||
                (e = hipMalloc((void **)&s.d_order, order_capacity(s.task_cap) * sizeof(uint32_t))) != hipSuccess ||
                (e = hipHostMalloc((void **)&s.h_out, s.task_cap * sizeof(bsw_result), hipHostMallocDefault)) != hipSuccess ||
                (e = hipMalloc((void **)&s.d_out, s.task_cap * sizeof(bsw_result))) != hipSuccess) { s.task_cap = 0; fail(&local, BSW_E_NOMEM, "staging: %s", hipGetErrorString(e)); return failed(BSW_E_NOMEM); }
        }
        packed_host ph;
        size_t w2 = 0;
        int rc = pack_tasks(&local, &p, tasks + base, cnt, threads, s.h_seq, s.seq_cap, ph, s.h_tasks, s.h_order, &w2);
        if (rc) return failed(rc);
        if ((e = hipMemcpyAsync(s.d_seq, s.h_seq, w2 * 8, hipMemcpyHostToDevice, st)) != hipSuccess ||
            (e = hipMemcpyAsync(s.d_tasks, s.h_tasks, cnt * sizeof(bsw_dtask), hipMemcpyHostToDevice, st)) != hipSuccess ||
            (e = hipMemcpyAsync(s.d_order, s.h_order, ((size_t)ph.plan.order_len + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, st)) != hipSuccess) { fail(&local, BSW_E_HIP, "H2D: %s", hipGetErrorString(e)); return failed(BSW_E_HIP); }
        rc = enqueue_batch(&local, dp, p.variant, s.d_seq, s.d_tasks, s.d_order, ph.plan, s.d_out, st, nullptr);
        if (rc) return failed(rc);
        if ((e = hipMemcpyAsync(s.h_out, s.d_out, cnt * sizeof(bsw_result), hipMemcpyDeviceToHost, st)) != hipSuccess ||
            (e = hipStreamSynchronize(st)) != hipSuccess) { fail(&local, BSW_E_HIP, "D2H: %s", hipGetErrorString(e)); return failed(BSW_E_HIP); }
        memcpy(out + base, s.h_out, cnt * sizeof(bsw_result));
    }
    return BSW_OK;
}

static int submit_pipeline(bsw_ctx *ctx, bsw_params p, const bsw_task *tasks, size_t n, bsw_result *out)
{
    bsw_dparams dp;
    int rc = check_params(ctx, &p, &dp);
    if (rc) return rc;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t chunk = ctx->cfg.chunk_tasks;
    const size_t nslots = ctx->streams.size();
    if (!ctx->slots) ctx->slots = new std::vector<slot_t>(nslots);
    const size_t nchunks = (n + chunk - 1) / chunk;
    const size_t nworkers = std::min(nslots, nchunks ? nchunks : 1);
    const int threads = std::max(1, ctx->cfg.pack_threads / (int)nworkers);
    std::atomic<int> abort_flag{0};
    std::vector<int> rcs(nworkers, 0);
    std::vector<std::string> errs(nworkers);
    std::vector<std::thread> th;
    for (size_t k = 1; k < nworkers; ++k)
        th.emplace_back([&, k]() { rcs[k] = slot_worker(ctx, p, dp, tasks, n, out, k, nslots, threads, abort_flag, errs[k]); });
    rcs[0] = slot_worker(ctx, p, dp, tasks, n, out, 0, nslots, threads, abort_flag, errs[0]);
    for (auto &t : th) t.join();
    for (size_t k = 0; k < nworkers; ++k)
        if (rcs[k]) { ctx->err = errs[k]; return rcs[k]; }
    return BSW_OK;
}

extern "C" int bsw_submit(bsw_ctx *ctx, const bsw_params *p, const bsw_task *tasks, size_t n, bsw_result *out)
{
    if (!ctx || !p || (!tasks && n) || (!out && n)) return fail(ctx, BSW_E_INVAL, "bsw_submit: NULL argument");
    if (ctx->worker_active) return fail(ctx, BSW_E_BUSY, "previous bsw_submit not waited for");
    bsw_dparams dp;
    int rc = check_params(ctx, p, &dp);
    if (rc) return rc;
    ctx->worker_active = true;
    ctx->worker_rc = 0;
    bsw_params pc = *p;
    ctx->worker = std::thread([ctx, pc, tasks, n, out]() { ctx->worker_rc = submit_pipeline(ctx, pc, tasks, n, out); });
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
extern "C" int bsw_extend_batch(bsw_ctx *ctx, const bsw_params *p, const bsw_ext_task *tasks, size_t n, bsw_ext *out)
{
    if (!ctx || !p || (!tasks && n) || (!out && n)) return fail(ctx, BSW_E_INVAL, "bsw_extend_batch: NULL argument");
    /* group by (w, end_bonus): each group is one pair-batch with only the right side populated,
     * one band try, clip penalties = end_bonus (they only feed max_ins/max_del here). */
    std::vector<uint32_t> idx(n);
    for (size_t i = 0; i < n; ++i) idx[i] = (uint32_t)i;
    std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) {
        if (tasks[a].w != tasks[b].w) return tasks[a].w < tasks[b].w;
        return tasks[a].end_bonus < tasks[b].end_bonus;
    });
    size_t g0 = 0;
    while (g0 < n) {
        size_t g1 = g0;
        while (g1 < n && tasks[idx[g1]].w == tasks[idx[g0]].w && tasks[idx[g1]].end_bonus == tasks[idx[g0]].end_bonus) ++g1;
        bsw_params pp = *p;
        pp.w = tasks[idx[g0]].w;
        pp.pen_clip3 = pp.pen_clip5 = tasks[idx[g0]].end_bonus;
        pp.max_band_try = 1;
        std::vector<bsw_task> pt(g1 - g0);
        for (size_t k = g0; k < g1; ++k) {
            const bsw_ext_task &e = tasks[idx[k]];
            bsw_task &t = pt[k - g0];
            memset(&t, 0, sizeof(t));
            if (e.qlen < 1) return fail(ctx, BSW_E_INVAL, "ext task %u: qlen must be >= 1", idx[k]);
            t.rquery = e.query; t.rtarget = e.target; t.rqlen = e.qlen; t.rtlen = e.tlen;
            t.h0 = e.h0; t.init_score = -1; t.tag = idx[k];
        }
        std::vector<bsw_result> res(g1 - g0);
        bsw_dev_batch *b = nullptr;
        int rc = bsw_upload(ctx, &pp, pt.data(), pt.size(), &b);
        if (rc) return rc;
        rc = bsw_run(ctx, b);
        if (!rc) rc = bsw_download(ctx, b, res.data());
        bsw_free_batch(ctx, b);
        if (rc) return rc;
        for (size_t k = g0; k < g1; ++k) {
            out[idx[k]] = res[k - g0].right;
            out[idx[k]].aw = tasks[idx[k]].w;
        }
        g0 = g1;
    }
    return BSW_OK;
}

/* ---- drop-in scalar ABI -------------------------------------------------------- */
static std::mutex g_mu;
static bsw_ctx *g_ctx = nullptr;
static std::atomic<int> g_variant{BSW_VARIANT_H};

extern "C" void bsw_set_default_variant(int variant) { g_variant = variant == BSW_VARIANT_M ? BSW_VARIANT_M : BSW_VARIANT_H; }

extern "C" int ksw_extend2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat,
                           int o_del, int e_del, int o_ins, int e_ins, int w, int end_bonus, int zdrop, int h0,
                           int *qle, int *tle, int *gtle, int *gscore, int *max_off)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_ctx) {
        bsw_config c;
        bsw_default_config(&c);
        const char *dv = getenv("BSW_DEVICE");
        if (dv) c.device = atoi(dv);
        int rc = bsw_create(&c, &g_ctx);
        if (rc) {
            fprintf(stderr, "ksw_extend2(libbwasw_mi355): cannot create GPU context (%d); no CPU fallback exists\n", rc);
            return -1;
        }
    }
    if (m != 5 || !mat || !query || (tlen > 0 && !target)) {
        fprintf(stderr, "ksw_extend2(libbwasw_mi355): unsupported arguments (m must be 5)\n");
        return -1;
    }
    bsw_params p;
    bsw_default_params(&p);
    memcpy(p.mat, mat, 25);
    p.o_del = o_del; p.e_del = e_del; p.o_ins = o_ins; p.e_ins = e_ins;
    p.zdrop = zdrop; p.variant = g_variant;
    bsw_ext_task t;
    memset(&t, 0, sizeof(t));
    t.query = query; t.target = target; t.qlen = qlen; t.tlen = tlen; t.w = w; t.end_bonus = end_bonus; t.h0 = h0;
    bsw_ext x;
    int rc = bsw_extend_batch(g_ctx, &p, &t, 1, &x);
    if (rc) {
        fprintf(stderr, "ksw_extend2(libbwasw_mi355): GPU path failed (%d): %s\n", rc, bsw_last_error(g_ctx));
        return -1;
    }
    if (qle) *qle = x.qle;
    if (tle) *tle = x.tle;
    if (gtle) *gtle = x.gtle;
    if (gscore) *gscore = x.gscore;
    if (max_off) *max_off = x.max_off;
    return x.score;
}

extern "C" int ksw_extend(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat,
                          int gapo, int gape, int w, int end_bonus, int zdrop, int h0,
                          int *qle, int *tle, int *gtle, int *gscore, int *max_off)
{
    return ksw_extend2(qlen, query, tlen, target, m, mat, gapo, gape, gapo, gape, w, end_bonus, zdrop, h0,
                       qle, tle, gtle, gscore, max_off);
}

/* ---- reference wire format end to end ----------------------------------------- */
extern "C" int bsw_refbatch_run(bsw_ctx *ctx, const uint32_t *in_words, uint32_t *out_words, int variant, int zdrop)
{
    if (!ctx || !in_words || !out_words) return fail(ctx, BSW_E_INVAL, "bsw_refbatch_run: NULL argument");
    const uint32_t n = in_words[2];
    if (n > BSW_REFBATCH_MAX_TASKS) return fail(ctx, BSW_E_LIMIT, "task batch announces %u tasks (> %d)", n, BSW_REFBATCH_MAX_TASKS);
    bsw_params p;
    std::vector<bsw_task> tasks(n ? n : 1);
    std::vector<uint8_t> seqbuf((size_t)BSW_REFBATCH_IN_WORDS * 8 + 64);
    int got = bsw_refbatch_decode(in_words, &p, tasks.data(), n, seqbuf.data(), seqbuf.size());
    if (got < 0) return fail(ctx, got, "malformed task batch");
    p.variant = variant; p.zdrop = zdrop;
    std::vector<bsw_result> res((size_t)got ? (size_t)got : 1);
    if (got) {
        bsw_dev_batch *b = nullptr;
        int rc = bsw_upload(ctx, &p, tasks.data(), (size_t)got, &b);
        if (rc) return rc;
        rc = bsw_run(ctx, b);
        if (!rc) rc = bsw_download(ctx, b, res.data());
        bsw_free_batch(ctx, b);
        if (rc) return rc;
    }
    memset(out_words, 0, BSW_REFBATCH_OUT_WORDS * sizeof(uint32_t));
    return bsw_refbatch_encode_results(res.data(), (size_t)got, out_words);
}
