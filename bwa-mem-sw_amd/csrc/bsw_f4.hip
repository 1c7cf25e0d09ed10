/*
 * bsw_f4.hip — hosts of the two other Smith-Waterman users of bwa: bsw_global_batch / ksw_global2 / ksw_global and bsw_align_batch / ksw_align2 / ksw_align
 * (part of the host side of libbwasw_mi355.so; shared types and the functions that cross files: bsw_internal.h)
 */
#include "bsw_internal.h"

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
    HIPCHK(e, bsw::launch_pack(st.d_raw.p, st.d_tasks.p, st.d_roff.p, 0u, (uint32_t)n, 0, nullptr, 0, nullptr, st.d_seq.p, nullptr, s));
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
        if (t.qlen > BSW_GLOBAL_MAX_QLEN || t.tlen > BSW_MAX_TLEN || t.w > BSW_MAX_TLEN) return fail(e, BSW_E_LIMIT, "global task %zu: beyond BSW_GLOBAL_MAX_QLEN/BSW_MAX_TLEN", i);
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
    HIPCHK(e, bsw::launch_pack(st.d_raw.p + RAW_FRONT, st.d_tasks.p, st.d_roff.p, 0u, (uint32_t)n, 0, nullptr, 0, nullptr, st.d_seq.p, nullptr, s));
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

