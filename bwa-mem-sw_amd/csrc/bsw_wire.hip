/*
 * bsw_wire.hip — the reference's wire format end to end: bsw_refbatch_submit / _wait / _run over 256 KiB task batches and 16 KiB result batches (bwa_mem_sw.v:163-170, task_parse.v:1931-1940)
 * (part of the host side of libbwasw_mi355.so; shared types and the functions that cross files: bsw_internal.h)
 */
#include "bsw_internal.h"

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
    ctx->ref_queue.push_back(refbatch_req{in_words, out_words, in_words[2]});
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
#define REFBATCH_GROUP_DEEP 64
#define REFBATCH_SLOTS 4
static int refbatch_enqueue(bsw_ctx *ctx, errs &e, size_t q0, size_t q1, int variant, int zdrop, stage_t &st, hipStream_t s, size_t *n_out)
{
    *n_out = 0;
    static const bool dbg = getenv("BSW_DEBUG_TIMING") != nullptr;
    auto tnow = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_a = dbg ? tnow() : 0;
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
    for (size_t q = q0; q < q1; ++q) n += ctx->ref_queue[q].nt;       /* (the count snapshot of bsw_refbatch_submit, never the header again) */
    if (n == 0) {
        for (size_t q = q0; q < q1; ++q) memset(ctx->ref_queue[q].out, 0, BSW_REFBATCH_OUT_WORDS * sizeof(uint32_t));
        return BSW_OK;
    }
    const size_t nb = q1 - q0, wire_words = nb * (size_t)BSW_REFBATCH_IN_WORDS;
    /* Task batches that sit in memory from bsw_host_alloc / bsw_host_register are DMA'd where they are — the reference's
     * batch manager reads the host's pinned workspace itself (batch_manager.v:745-773) — instead of being copied into the
     * slot's pinned staging first (16 MB of memcpy per 64 batches: 0.6 ms of a group's 0.8 ms on the host) */
    bool direct = true;
    for (size_t q = q0; q < q1 && direct; ++q) direct = is_registered(ctx->ref_queue[q].in, BSW_REFBATCH_IN_WORDS * sizeof(uint32_t));
    hipError_t he;
    if ((he = st.h_tasks.reserve(n + 1)) != hipSuccess || (he = st.h_woff.reserve(n + 1)) != hipSuccess || (!direct && (he = st.h_raw.reserve(wire_words * 4 + RAW_SLACK)) != hipSuccess))
        return fail(e, BSW_E_NOMEM, "pinned staging: %s", hipGetErrorString(he));
    /* The batches themselves start crossing PCIe BEFORE the headers are parsed when they can be DMA'd in place: 16 MB per 64
     * batches = 0.3 ms that used to follow the 0.35 ms parse now runs beside it (profiles/r5/wire_format_timeline.txt) */
    if ((he = st.d_raw.reserve(wire_words * 4 + RAW_SLACK)) != hipSuccess) return fail(e, BSW_E_NOMEM, "device staging: %s", hipGetErrorString(he));
    if (direct)
        for (size_t q = q0; q < q1;) {                 /* one DMA per run of batches that lie back to back in the caller's memory */
            size_t r1 = q + 1;
            /* (a run ends where the next batch is not adjacent OR the span would leave one registered range: two registrations
             * that happen to touch are two DMAs) */
            while (r1 < q1 && ctx->ref_queue[r1].in == ctx->ref_queue[r1 - 1].in + BSW_REFBATCH_IN_WORDS &&
                   is_registered(ctx->ref_queue[q].in, (r1 + 1 - q) * (size_t)BSW_REFBATCH_IN_WORDS * sizeof(uint32_t))) ++r1;
            HIPCHK(e, hipMemcpyAsync(st.d_raw.p + (q - q0) * (size_t)BSW_REFBATCH_IN_WORDS * 4, ctx->ref_queue[q].in,
                                     (r1 - q) * (size_t)BSW_REFBATCH_IN_WORDS * 4, hipMemcpyHostToDevice, s));
            q = r1;
        }
    /* headers -> task records (host: 8 words per task) */
    chunk_info ci;
    rc = fill_binparams(e, &p, ctx->cfg.kernel, ci.bp);
    if (rc) return rc;
    bsw_binparams &bp = ci.bp;
    uint32_t cw_all[BSW_MAX_WAVE_CLASSES] = {0}, cw[BSW_MAX_WAVE_CLASSES] = {0}, cw16[BSW_MAX_WAVE_CLASSES] = {0};
    uint32_t cl[BSW_MAX_LANE_CLASSES] = {0}, cr[BSW_MAX_LANE_CLASSES] = {0}, n_lane = 0, n16 = 0;
    uint64_t acc = 0, w8l = 0, w8r = 0;
    int h0_lo = INT_MAX, h0_hi = 0;                /* h0 range of the left sides that go to lane classes (bsw_h0_bucket) */
    /* pass 1: where every batch's tasks and sequence words start (the batches' word counts side by side, then a prefix sum) */
    std::vector<uint64_t> wbase(nb + 1, 0), tbase(nb + 1, 0);
    /* ONE thread walks a group of up to 128 batches faster than several do (64 batches = 52 k headers: 0.36 ms on one thread,
     * 0.95 / 0.50 / 0.40 ms on 2 / 4 / 8 freshly started ones, profiles/r5/wire_format_group_sweep.txt) — and then needs no
     * first pass at all: the offsets run along */
    const size_t nth = nb <= 128 ? 1 : std::max<size_t>(1, std::min<size_t>((size_t)ctx->cfg.pack_threads, nb / 32));
    if (nth > 1) {
        auto count = [&](size_t t) {
            for (size_t q = q0 + t; q < q1; q += nth) {
                const uint32_t *W = ctx->ref_queue[q].in;
                const uint32_t nt = ctx->ref_queue[q].nt;
                uint64_t words = 0;
                for (uint32_t i = 0; i < nt && 8 + 8 * (uint64_t)i + 1 < BSW_REFBATCH_IN_WORDS; ++i) {
                    const uint32_t *H = &W[8 + 8 * i];
                    const int lq = (int)(H[0] & 0xff), lt = (int)((H[0] >> 16) & 0x7ff), rq = (int)(H[1] & 0xff), rt = (int)((H[1] >> 16) & 0x7ff);
                    words += (lq ? nwords(lq) + nwords(lt) : 0) + (rq ? nwords(rq) + nwords(rt) : 0);
                }
                wbase[q - q0 + 1] = words;
                tbase[q - q0 + 1] = nt;
            }
        };
        std::vector<std::thread> th;
        for (size_t t = 1; t < nth; ++t) th.emplace_back(count, t);
        count(0);
        for (auto &x : th) x.join();
        for (size_t k = 1; k <= nb; ++k) { wbase[k] += wbase[k - 1]; tbase[k] += tbase[k - 1]; }
    }
    const double t_b = dbg ? tnow() : 0;
    /* pass 2 (parallel over batches): records, class counts, and the batch itself into pinned staging */
    struct part { uint32_t cw_all[BSW_MAX_WAVE_CLASSES] = {0}, cw[BSW_MAX_WAVE_CLASSES] = {0}, cw16[BSW_MAX_WAVE_CLASSES] = {0}, cl[BSW_MAX_LANE_CLASSES] = {0}, cr[BSW_MAX_LANE_CLASSES] = {0}, n_lane = 0, n16 = 0; uint64_t w8l = 0, w8r = 0; int h0_lo = INT_MAX, h0_hi = 0; int rc = 0; errs e; };
    std::vector<part> parts(nth);
    /* class of a side / a seed by query length, looked up instead of searched per task (wire lengths are 8-bit fields) */
    uint8_t wcls[256], lcls8[256], lcls16[256];
    for (int q = 0; q < 256; ++q) {
        wcls[q] = (uint8_t)bsw_wave_class_of(&bp, q);
        lcls8[q] = (uint8_t)bsw_side_lane_class(&bp, 8, q);
        lcls16[q] = (uint8_t)bsw_side_lane_class(&bp, 16, q);
    }
    auto parse = [&](size_t t) {
        part &pt = parts[t];
        for (size_t q = q0 + t; q < q1; q += nth) {
            const uint32_t *W = ctx->ref_queue[q].in;
            const uint32_t nt = ctx->ref_queue[q].nt;
            uint64_t a2 = wbase[q - q0];
            size_t ti = (size_t)tbase[q - q0];
            if (nth == 1) tbase[q - q0 + 1] = tbase[q - q0] + nt;
            const int64_t base = nt ? (int64_t)(8 + 8 * nt) - (int64_t)W[8 + 2] : 0;
            for (uint32_t i = 0; i < nt; ++i, ++ti) {
                const uint32_t *H = &W[8 + 8 * i];
                bsw_dtask d;                          /* built here, stored to the pinned staging in one go below */
                bsw_wireoff wo;
                memset(&d, 0, sizeof(d));
                const int lq = (int)(H[0] & 0xff), lt = (int)((H[0] >> 16) & 0x7ff), rq = (int)(H[1] & 0xff), rt = (int)((H[1] >> 16) & 0x7ff);
                const int64_t pos = base + (int64_t)H[2];
                if (pos < 8 + 8 * (int64_t)nt || pos + (lq + rq + lt + rt + 7) / 8 > BSW_REFBATCH_IN_WORDS) {
                    pt.rc = fail(pt.e, BSW_E_INVAL, "malformed task batch (task %u: data position)", i);
                    return;
                }
                const int h0 = (int)(H[4] & 0xff);
                if (h0 <= 0) { pt.rc = fail(pt.e, BSW_E_INVAL, "task batch: task %u has h0 <= 0", i); return; }
                wo.nib = (uint32_t)(((uint64_t)(q - q0) * BSW_REFBATCH_IN_WORDS + (uint64_t)pos) * 8u);
                wo.out_word = (uint32_t)((q - q0) * (size_t)BSW_REFBATCH_OUT_WORDS + 5u * i);
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
                const int wc = wcls[std::max(lq, rq)];
                ++pt.cw_all[wc];
                const int bits = bsw_seed_lane_bits(&bp, lq, rq, h0);
                if (!bits) ++pt.cw[wc];
                else {
                    const uint8_t *lc = bits == 8 ? lcls8 : lcls16;
                    ++pt.n_lane;
                    if (bits == 16) { ++pt.n16; ++pt.cw16[wc]; }
                    if (lq) { ++pt.cl[lc[lq]]; if (bits == 8) { pt.h0_lo = std::min(pt.h0_lo, h0); pt.h0_hi = std::max(pt.h0_hi, h0); } }
                    if (rq) ++pt.cr[lc[rq]];
                    if (bits == 8) { pt.w8l += (uint64_t)lq; pt.w8r += (uint64_t)rq; }
                }
                st.h_tasks.p[ti] = d;
                st.h_woff.p[ti] = wo;
            }
            if (nth == 1) wbase[q - q0 + 1] = a2;
            if (!direct) memcpy((uint32_t *)st.h_raw.p + (q - q0) * (size_t)BSW_REFBATCH_IN_WORDS, W, BSW_REFBATCH_IN_WORDS * sizeof(uint32_t));
        }
    };
    {
        std::vector<std::thread> th;
        for (size_t t = 1; t < nth; ++t) th.emplace_back(parse, t);
        parse(0);
        for (auto &x : th) x.join();
    }
    const double t_c = dbg ? tnow() : 0;
    acc = wbase[nb];
    for (const part &pt : parts) {
        if (pt.rc) { e = pt.e; return pt.rc; }
        for (int c = 0; c < BSW_MAX_WAVE_CLASSES; ++c) { cw_all[c] += pt.cw_all[c]; cw[c] += pt.cw[c]; cw16[c] += pt.cw16[c]; }
        for (int c = 0; c < BSW_MAX_LANE_CLASSES; ++c) { cl[c] += pt.cl[c]; cr[c] += pt.cr[c]; }
        n_lane += pt.n_lane; n16 += pt.n16; w8l += pt.w8l; w8r += pt.w8r;
        h0_lo = std::min(h0_lo, pt.h0_lo); h0_hi = std::max(h0_hi, pt.h0_hi);
    }
    const bool packed_ok = bsw::lane_class_finishes(0, dp, variant);
    const bool group = decide_lane_mode(ctx->cfg.kernel, packed_ok, bp, n_lane, n16, cl, cr, cw, cw16, nullptr, w8l, w8r);
    if (bp.lane_on && narrow_foldable(bp)) narrow_fold(bp, cl, cr, nullptr);     /* (wire-format groups: one launch per side) */
    if (!bp.lane_on) { memcpy(cw, cw_all, sizeof(cw)); memset(cl, 0, sizeof(cl)); memset(cr, 0, sizeof(cr)); n_lane = 0; }
    const int fused_cls = fuse_lists(bp, ctx->cfg.kernel, group, packed_ok, group ? n_lane : n_lane - n16, cl, cr);
    if (bp.lane_on && h0_hi >= h0_lo) bsw_set_h0_buckets(&bp, h0_lo, h0_hi);
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
    plan_fused(pl, bp, fused_cls, group, cl);
    plan_nsplit(pl, bp, nsplit_pays(ctx->cfg.kernel, bp, packed_ok, group ? n_lane : n_lane - n16, false), n_lane);
    /* (pl.dep stays all ones: the wire-format groups run their classes on one stream) */
    memcpy(bp.wave_start, pl.wave_start, sizeof(bp.wave_start));
    bp.lane_all_off = pl.lane_all_off;
    memcpy(bp.laneL_off, pl.laneL_off, sizeof(bp.laneL_off));
    memcpy(bp.laneR_off, pl.laneR_off, sizeof(bp.laneR_off));
    /* device: wire batches -> seq, bins, DP kernels, results */
    const size_t wout_words = nb * (size_t)BSW_REFBATCH_OUT_WORDS;
    if ((he = st.d_seq.reserve((size_t)acc + 4)) != hipSuccess ||
        (he = st.d_tasks.reserve(n + 1)) != hipSuccess || (he = st.d_woff.reserve(n + 1)) != hipSuccess ||
        (he = st.d_order.reserve(order_capacity(n))) != hipSuccess || (he = st.d_bins.reserve(BSW_BIN_WORDS)) != hipSuccess ||
        (he = st.d_keys.reserve(n + 1)) != hipSuccess ||
        (he = st.d_out.reserve(n + 1)) != hipSuccess || (he = st.d_wout.reserve(wout_words)) != hipSuccess)
        return fail(e, BSW_E_NOMEM, "device staging: %s", hipGetErrorString(he));
    if ((he = st.h_wout.reserve(wout_words)) != hipSuccess) return fail(e, BSW_E_NOMEM, "pinned staging: %s", hipGetErrorString(he));
    if (!direct) HIPCHK(e, hipMemcpyAsync(st.d_raw.p, st.h_raw.p, wire_words * 4, hipMemcpyHostToDevice, s));
    HIPCHK(e, hipMemcpyAsync(st.d_tasks.p, st.h_tasks.p, n * sizeof(bsw_dtask), hipMemcpyHostToDevice, s));
    HIPCHK(e, hipMemcpyAsync(st.d_woff.p, st.h_woff.p, n * sizeof(bsw_wireoff), hipMemcpyHostToDevice, s));
    HIPCHK(e, bsw::launch_wire_pack((const uint32_t *)st.d_raw.p, st.d_tasks.p, st.d_woff.p, (uint32_t)n, st.d_seq.p, s));
    HIPCHK(e, bsw::launch_bin(bp, st.d_seq.p, nullptr, st.d_tasks.p, (uint32_t)n, st.d_bins.p, st.d_keys.p, st.d_order.p, s));
    rc = enqueue_batch(e, dp, variant, st.d_seq.p, st.d_tasks.p, st.d_order.p, pl, st.d_out.p, s, nullptr);
    if (rc) return rc;
    /* the 16 KiB result batches are written on the device (five words of a 96-byte record per task: a fifth of the bytes over
     * PCIe, no encoding pass on the host).  Their DMA is issued by refbatch_collect_issue — once the kernels are done, or
     * when no further group follows: a copy queued now would sit in its DMA engine's ring until then and hold up the next
     * group's input copies queued behind it */
    HIPCHK(e, bsw::launch_wire_results(st.d_out.p, st.d_woff.p, (uint32_t)n, st.d_wout.p, wout_words, s));
    *n_out = n;
    if (dbg) fprintf(stderr, "[bsw] wire:   enqueue: reserve + pass 1 %.3f ms, pass 2 (%zu threads) %.3f, plan + device enqueue %.3f\n", t_b - t_a, nth, t_c - t_b, tnow() - t_c);
    return BSW_OK;
}

/* do the group's result batches all lie in memory the GPU can DMA into (bsw_host_alloc / bsw_host_register)? */
static bool outs_registered(bsw_ctx *ctx, size_t q0, size_t q1)
{
    for (size_t q = q0; q < q1; ++q)
        if (!is_registered(ctx->ref_queue[q].out, BSW_REFBATCH_OUT_WORDS * sizeof(uint32_t))) return false;
    return true;
}

/* queue the DMA of an enqueued group's result batches behind its kernels (wait_kernels: only once they are done, for a
 * group that other groups' input copies still follow).  Result batches in registered memory are written where they are
 * (one DMA per run that lies back to back) — the reference's manager writes its RBBs into the host's workspace itself
 * (rbb.v:150-166) —, others go through pinned staging and are copied out by refbatch_collect.  *direct_out says which. */
static int refbatch_collect_issue(bsw_ctx *ctx, size_t q0, size_t q1, stage_t &st, hipStream_t s, hipEvent_t ev, bool wait_kernels, bool *direct_out)
{
    errs &e = ctx->err;
    if (wait_kernels) {
        const int rc0 = sync_stream(ctx, e, s, ev);
        if (rc0) return rc0;
    }
    const size_t bw = (size_t)BSW_REFBATCH_OUT_WORDS;
    *direct_out = outs_registered(ctx, q0, q1);
    if (!*direct_out) {
        HIPCHK(e, hipMemcpyAsync(st.h_wout.p, st.d_wout.p, (q1 - q0) * bw * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        return BSW_OK;
    }
    for (size_t q = q0; q < q1;) {
        size_t r1 = q + 1;
        while (r1 < q1 && ctx->ref_queue[r1].out == ctx->ref_queue[r1 - 1].out + bw &&
               is_registered(ctx->ref_queue[q].out, (r1 + 1 - q) * (size_t)bw * sizeof(uint32_t))) ++r1;
        HIPCHK(e, hipMemcpyAsync(ctx->ref_queue[q].out, st.d_wout.p + (q - q0) * bw, (r1 - q) * bw * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        q = r1;
    }
    return BSW_OK;
}

/* wait for a group's result batches and hand them to the caller's buffers */
static int refbatch_collect(bsw_ctx *ctx, size_t q0, size_t q1, stage_t &st, hipStream_t s, hipEvent_t ev, bool direct_out)
{
    errs &e = ctx->err;
    const int rc = sync_stream(ctx, e, s, ev);
    if (rc) return rc;
    if (!direct_out)
        for (size_t q = q0; q < q1; ++q)
            memcpy(ctx->ref_queue[q].out, st.h_wout.p + (q - q0) * (size_t)BSW_REFBATCH_OUT_WORDS, BSW_REFBATCH_OUT_WORDS * sizeof(uint32_t));
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
    /* Group size.  A deep queue (>= 96 batches = 79 k seeds) is cut into TWO groups of at least 64 batches: 52 k PE seeds are
     * past the point where the two-seeds-per-lane kernels beat the general one (48 k mixed seeds,
     * profiles/r5/crossover_mixed_bins.json), and two groups in flight overlap each other's launch floors.  Not more: a group's
     * chain (left launch, right launch: one wave's lifetime each) is 2.2 ms however few seeds it holds, and four groups of 64
     * took 6.3 ms per 256 batches where two of 128 take 4.6 and three of 86 take 4.5 — the fourth slot stream shares a
     * hardware queue (profiles/r5/wire_format_timeline.txt; r3/e2e_hw_queues.txt).  A shallow queue keeps groups of 16 on the
     * general kernel, which finishes 13 k seeds sooner than two lane launches.  BSW_REFBATCH_GROUP overrides (measurements). */
    static const size_t grp_tune = getenv("BSW_REFBATCH_GROUP") ? (size_t)std::max(1, atoi(getenv("BSW_REFBATCH_GROUP"))) : 0;
    /* (round 6) a queue of 24 - 95 batches runs as groups of 24 = 19.7 k seeds: that is bsw_lane2g_kernel's range with both sides
     * of a seed in one launch (a seed pair per group of eight lanes: 0.63 ms per group where two lane launches take 2.0 and the
     * general kernels 0.85), and three or four such groups overlap their DMAs with each other's kernels: 32 queued batches
     * 17.5 -> 20.3 M seeds/s, 48: 21.5 -> 25.7, 64: 19.2 (round 5's rule) -> 29.1 (profiles/r6/wire_format_group_sweep.txt;
     * groups of 48, one launch per side, were the first version of this rule: 26.2 at 64).  At most three groups: a fourth shares a
     * hardware queue (96 batches as four groups of 24: 27.5 M, as three of 32: 33.1).
     * A deep queue (>= 96 batches) now goes in groups of up to 128 batches = 105 k seeds: the lane kernels run both sides of such
     * a group in ONE launch (bsw_lane2_kernel's fused instantiation, 1.3 ms where two launches took 2.0), so one big group beats
     * two halves in flight: 96 batches 28.3 -> 35.1 M seeds/s, 128: 36.7 -> 40.7, 192: 45.9 -> 48.9, 256: 53 (second table of
     * profiles/r6/wire_format_group_sweep.txt). */
    const size_t grp_env = grp_tune ? grp_tune : (nq >= 96 ? std::min<size_t>(nq, 2 * REFBATCH_GROUP_DEEP) : nq >= 24 ? std::max<size_t>(24, (nq + 2) / 3) : (size_t)REFBATCH_GROUP);
    const size_t NS = std::max<size_t>(1, std::min<size_t>(REFBATCH_SLOTS, dev.slots.size()));
    const size_t slot_of[REFBATCH_SLOTS] = {0, 1, 2, 3};
    struct flight { bool active = false, issued = false, direct_out = false; size_t q0 = 0, q1 = 0; } fl[REFBATCH_SLOTS];
    static const bool dbg = getenv("BSW_DEBUG_TIMING") != nullptr;
    auto tnow = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_start = tnow();
    auto issue = [&](size_t sl, bool wait_kernels) -> int {
        if (!fl[sl].active || fl[sl].issued) return BSW_OK;
        fl[sl].issued = true;
        return refbatch_collect_issue(ctx, fl[sl].q0, fl[sl].q1, dev.slots[slot_of[sl]], dev.streams[slot_of[sl]], dev.events[slot_of[sl]], wait_kernels, &fl[sl].direct_out);
    };
    auto collect = [&](size_t sl) -> int {
        if (!fl[sl].active) return BSW_OK;
        const double t0 = tnow();
        int r = issue(sl, true);
        fl[sl].active = false;
        if (!r) r = refbatch_collect(ctx, fl[sl].q0, fl[sl].q1, dev.slots[slot_of[sl]], dev.streams[slot_of[sl]], dev.events[slot_of[sl]], fl[sl].direct_out);
        if (dbg) fprintf(stderr, "[bsw] wire: collect slot %zu batches [%zu,%zu): +%.3f .. +%.3f ms\n", sl, fl[sl].q0, fl[sl].q1, t0 - t_start, tnow() - t_start);
        return r;
    };
    auto drain = [&]() { for (size_t sl = 0; sl < NS; ++sl) if (fl[sl].active) { errs quiet; (void)sync_stream(ctx, quiet, dev.streams[slot_of[sl]], dev.events[slot_of[sl]]); fl[sl].active = false; } };
    /* the groups, in queue order */
    struct grp_t { size_t q0, q1; };
    std::vector<grp_t> groups;
    for (size_t q0 = 0; q0 < nq;) {
        size_t q1 = q0 + 1;
        while (q1 < nq && q1 - q0 < grp_env && ctx->ref_queue[q1].in[0] == ctx->ref_queue[q0].in[0] && ctx->ref_queue[q1].in[1] == ctx->ref_queue[q0].in[1]) ++q1;
        if (nq - q1 < grp_env / 2)                                 /* no runt group at the end of a run */
            while (q1 < nq && ctx->ref_queue[q1].in[0] == ctx->ref_queue[q0].in[0] && ctx->ref_queue[q1].in[1] == ctx->ref_queue[q0].in[1]) ++q1;
        groups.push_back(grp_t{q0, q1});
        q0 = q1;
    }
    size_t k = 0;
    /* Big groups that all find a free slot are parsed and enqueued SIDE BY SIDE, one host thread each: a group's chain of
     * launches cannot start before its headers are parsed (0.36 ms per 64 batches), and parsed one after the other the last
     * group of a 128 / 256-batch queue started 0.4 / 1.2 ms after the first (profiles/r5/wire_format_timeline.txt).  Small
     * groups (a shallow queue's 16 batches: 0.09 ms of headers) are not worth a thread start. */
    if (groups.size() >= 2 && groups.size() <= NS && groups[0].q1 - groups[0].q0 >= 32) {
        const size_t G = groups.size();
        std::vector<errs> ge(G);
        std::vector<int> grc(G, 0);
        std::vector<size_t> gn(G, 0);
        const int device = ctx->device0();
        auto work = [&](size_t g, bool helper) {
            if (helper && hipSetDevice(device) != hipSuccess) { grc[g] = fail(ge[g], BSW_E_HIP, "hipSetDevice"); return; }
            const double te = tnow();
            grc[g] = refbatch_enqueue(ctx, ge[g], groups[g].q0, groups[g].q1, variant, zdrop, dev.slots[slot_of[g]], dev.streams[slot_of[g]], &gn[g]);
            if (dbg) fprintf(stderr, "[bsw] wire: enqueue slot %zu batches [%zu,%zu) %zu tasks: +%.3f .. +%.3f ms (side by side)\n", g, groups[g].q0, groups[g].q1, gn[g], te - t_start, tnow() - t_start);
        };
        {
            std::vector<std::thread> th;
            for (size_t g = 1; g < G; ++g) th.emplace_back(work, g, true);
            work(0, false);
            for (auto &x : th) x.join();
        }
        for (size_t g = 0; g < G; ++g)
            if (grc[g] == 0 && gn[g]) { fl[g].active = true; fl[g].issued = false; fl[g].q0 = groups[g].q0; fl[g].q1 = groups[g].q1; }
        for (size_t g = 0; g < G; ++g)
            if (grc[g]) {                                          /* (as below: every stream that may hold work is drained first) */
                for (size_t h = 0; h < G; ++h) { errs quiet; (void)sync_stream(ctx, quiet, dev.streams[slot_of[h]], dev.events[slot_of[h]]); fl[h].active = false; }
                e = ge[g];
                ctx->ref_queue.clear();
                return grc[g];
            }
        k = G;
    }
    for (size_t g = k; g < groups.size(); ++g) {
        const size_t q0 = groups[g].q0, q1 = groups[g].q1;
        const size_t sl = k++ % NS;
        rc = collect(sl);
        size_t n_enq = 0;
        const double te = tnow();
        if (!rc) rc = refbatch_enqueue(ctx, e, q0, q1, variant, zdrop, dev.slots[slot_of[sl]], dev.streams[slot_of[sl]], &n_enq);
        if (dbg) fprintf(stderr, "[bsw] wire: enqueue slot %zu batches [%zu,%zu) %zu tasks: +%.3f .. +%.3f ms\n", sl, q0, q1, n_enq, te - t_start, tnow() - t_start);
        if (rc) {
            /* a failure half-way through refbatch_enqueue leaves copies out of the queued batches / kernels on this slot's
             * stream with fl[sl].active still false: drain that stream too before the caller gets its buffers back */
            { errs quiet; (void)sync_stream(ctx, quiet, dev.streams[slot_of[sl]], dev.events[slot_of[sl]]); }
            drain(); ctx->ref_queue.clear(); return rc;
        }
        if (n_enq) { fl[sl].active = true; fl[sl].issued = false; fl[sl].q0 = q0; fl[sl].q1 = q1; }
    }
    /* nothing follows: every group's result DMA is queued behind its kernels now, no host round trip in between */
    for (size_t sl = 0; sl < NS; ++sl) {
        rc = issue((k + sl) % NS, false);
        if (rc) { drain(); ctx->ref_queue.clear(); return rc; }
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

