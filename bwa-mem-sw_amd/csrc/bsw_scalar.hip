/*
 * bsw_scalar.hip — batched plain ksw_extend2 (bsw_extend_batch) and the drop-in scalar entry points ksw_extend / ksw_extend2 with the leader/follower queue they share with ksw_align2 / ksw_global2 (software interface of sw_pe_array_sw_extend.v:96-123)
 * (part of the host side of libbwasw_mi355.so; shared types and the functions that cross files: bsw_internal.h)
 */
#include "bsw_internal.h"

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
    const size_t chunk = ctx->cfg.chunk_tasks ? ctx->cfg.chunk_tasks : 131072;
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
BSW_LOCAL void scalar_call(scalar_req &req)
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

