/*
 * bsw_batch.hip — the batch manager: host pass over a chunk (validate, lay out, count per kernel class), device staging (DMA, pack, bin), kernel launches, batch plan export, device-resident batches and reference, the small-batch path, the streaming slot pipeline behind bsw_submit / bsw_submit_packed / bsw_submit_ref (batch_manager.v:358-739, tbb.v, rbb.v)
 * (part of the host side of libbwasw_mi355.so; shared types and the functions that cross files: bsw_internal.h)
 */
#include "bsw_internal.h"

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

static inline uint64_t thread_cpu_ns()
{
    struct timespec ts;
    clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
    return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}

BSW_LOCAL size_t order_capacity(size_t n) { return 5 * n + 32; }   /* upper bound of plan.order_len (with the N list of bsw_binparams.nsplit) + the 3 + BSW_MAX_WAVE_CLASSES counters behind it */

/* ---- host pass over a chunk: validate, lay out, count per class ------------------------------ */


BSW_LOCAL int fill_binparams(errs &e, const bsw_params *p, int kern, bsw_binparams &bp)
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
BSW_LOCAL bool narrow_foldable(const bsw_binparams &bp)
{
    return bp.n_lane >= 2 && bp.lane_bits[0] == bp.lane_bits[1] && bp.lane_cols[0] > 0 && bp.lane_cols[0] < bp.lane_cols[1];
}
static double narrow_min_share()
{
    static const double v = getenv("BSW_NARROW_SHARE") ? atof(getenv("BSW_NARROW_SHARE")) : NARROW_MIN_SHARE;
    return v;
}
/* fold class 0 into class 1: counts, dependency bits, and the class table the device sorts with */
BSW_LOCAL void narrow_fold(bsw_binparams &bp, uint32_t *cl, uint32_t *cr, uint8_t *dep)
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

/* work8_l / work8_r: the summed query lengths of the chunk's 8-bit left / right lane sides */
BSW_LOCAL bool decide_lane_mode(int kern, bool group_ok, bsw_binparams &bp, uint32_t &n_lane, uint32_t n16, uint32_t *cl, uint32_t *cr,
                                uint32_t *cw, const uint32_t *cw16, uint8_t *dep, uint64_t work8_l, uint64_t work8_r, bool streaming)
{
    /* BSW_GROUP=0: never the group kernel (the lane kernels then start at round 5's seed count); 1: the group kernel for every
     * chunk with 8-bit lane seeds (tests, measurements) */
    static const int genv = getenv("BSW_GROUP") ? atoi(getenv("BSW_GROUP")) : -1;
    if (!bp.lane_on) return false;
    const uint32_t n8 = n_lane - n16;
    const uint64_t work = std::max(work8_l, work8_r);
    /* what the group kernel's launch holds: one side's seeds — or, for a chunk that fuse_group_lists will fuse, both sides of
     * every seed (one wave lifetime for the pair: the general kernels lose from ~13 k PE seeds, profiles/r6/crossover_group.json) */
    static const int fenv = getenv("BSW_GROUP_FUSE") ? atoi(getenv("BSW_GROUP_FUSE")) : -1;
    int cmax8 = -1;
    for (int c = 0; c < bp.n_lane && c < BSW_MAX_LANE_CLASSES; ++c)
        if (bp.lane_bits[c] == 8 && (cl[c] || cr[c])) cmax8 = c;
    const bool wide = cmax8 >= 0 && bp.lane_cols[cmax8] > 136;         /* sides beyond 136 columns (250 bp reads): the lane kernels have no fused form */
    const bool fusable = fenv != 0 && work8_l && work8_r && (fenv == 1 || n8 <= group_fuse_max(wide));
    const uint64_t work_g = fusable ? work8_l + work8_r : work;
    /* ... and past the group kernel's fused range the LANE kernels run such a chunk in one launch (bsw_lane2_kernel's fused
     * instantiation, 136 columns: 1.2 - 1.7 ms for 65 k - 262 k PE seeds where two lane launches take 2.0 and two group launches
     * 1.4 - 4.0, profiles/r6/crossover_group.json): lane bins from LANE_WORK_MIN bases over both sides */
    static const int lenv = getenv("BSW_LANE_FUSE") ? atoi(getenv("BSW_LANE_FUSE")) : -1;
    const bool lane_fusable = group_ok && lenv != 0 && !fusable && (!streaming || lenv == 1) && work8_l && work8_r && n8 <= LANE_FUSE_MAX && cmax8 >= 0 && !wide;
    bool group = false;
    if (genv == 1) group = group_ok && n8 > 0 && kern != BSW_KERNEL_WAVE;
    else if (kern == BSW_KERNEL_AUTO) {
        /* (a chunk the group kernel runs fused stays with it whatever its work: 250 bp reads, 49 k seeds 2.8 ms fused against 5.9
         * for the lane kernels' four chained launches, profiles/r6/crossover_group_250bp.json) */
        const bool lane = genv == 0 || !group_ok ? lane_bins_pay(n_lane, cl, cr)
                                                 : ((!fusable && work >= LANE_WORK_MIN) || lane_bins_pay(n16, cl, cr) || (lane_fusable && work8_l + work8_r >= LANE_WORK_MIN));
        if (!lane) {
            group = genv != 0 && group_ok && work_g >= (wide ? GROUP_WORK_MIN_WIDE : GROUP_WORK_MIN);
            if (!group) { bp.lane_on = 0; return false; }
        }
    }
    if (!group) return false;
    /* the group kernel serves the 8-bit classes; the chunk's 16-bit seeds (scores beyond 255) go to the general kernel */
    bp.cols16 = 0;
    for (int c = 0; c < BSW_MAX_LANE_CLASSES; ++c)
        if (c < bp.n_lane && bp.lane_bits[c] == 16) { cl[c] = cr[c] = 0; if (dep) dep[c] = 0; }
    for (int c = 0; c < BSW_MAX_WAVE_CLASSES; ++c) cw[c] += cw16[c];
    n_lane -= n16;
    return true;
}

/* See bsw_internal.h.  A seed's two sides add up to about one read length whatever the split, so a wavefront that runs the left
 * sides of its seeds and then their right sides lives little longer than ONE launch of the longest sides does, and the chunk
 * takes one wave lifetime instead of two — what a chunk that does not fill the machine (GROUP_FUSE_MAX seeds at 16 per
 * wavefront, LANE_FUSE_MAX at 128) is bound by.  Larger chunks are bound by throughput, and there a right-side list sorted by
 * its own length packs better than the left list's order.  BSW_GROUP_FUSE / BSW_LANE_FUSE = 0 / 1: never / whenever the chunk
 * has both kinds of sides (measurements, tests). */
BSW_LOCAL int fuse_lists(bsw_binparams &bp, int kern, bool group, bool packed_ok, uint32_t n8, uint32_t *cl, uint32_t *cr, bool streaming)
{
    static const int genv = getenv("BSW_GROUP_FUSE") ? atoi(getenv("BSW_GROUP_FUSE")) : -1;
    static const int lenv = getenv("BSW_LANE_FUSE") ? atoi(getenv("BSW_LANE_FUSE")) : -1;
    const int fenv = group ? genv : lenv;
    if (!bp.lane_on || !packed_ok || fenv == 0) return -1;
    if (!group && fenv != 1 && (kern != BSW_KERNEL_AUTO || streaming)) return -1;      /* (forced lane bins keep a list per side: BSW_KERNEL_LANE is what the tests of that path use) */
    int cmax = -1;
    uint32_t nl = 0, nr = 0;
    for (int c = 0; c < bp.n_lane && c < BSW_MAX_LANE_CLASSES; ++c) {
        if (bp.lane_bits[c] != 8) continue;
        if (cl[c] || cr[c]) cmax = c;
        nl += cl[c]; nr += cr[c];
    }
    if (fenv != 1 && n8 > (group ? group_fuse_max(cmax >= 0 && bp.lane_cols[cmax] > 136) : LANE_FUSE_MAX)) return -1;
    const int c0 = bsw_side_lane_class(&bp, 8, 0);
    if (cmax < 0 || c0 < 0 || !nl || !nr || nl > n8) return -1;       /* (a one-sided chunk has nothing to fuse) */
    if (bp.lane_cols[cmax] > (group ? 256 : 136)) return -1;           /* (the lane kernels: bsw_lane2_kernel<17>; the 232-column class has no fused form) */
    cl[c0] += n8 - nl;                                   /* the seeds without a left side: query length 0 */
    for (int c = 0; c < bp.n_lane && c < BSW_MAX_LANE_CLASSES; ++c)
        if (bp.lane_bits[c] == 8) cr[c] = 0;
    bp.fused = group ? 1 : 2;                            /* (the two differ in where the queries with an N go: bsw_stage_kernel.hip, seed_keys) */
    return cmax;
}

/* ... and where that launch finds its seeds, once the plan's lists are laid out */
BSW_LOCAL void plan_fused(batch_plan &pl, const bsw_binparams &bp, int fused_cls, bool group, const uint32_t *cl)
{
    pl.lane_group = group ? (fused_cls >= 0 ? 2 : 1) : 0;
    pl.fused_cls = fused_cls;
    pl.fused_off = pl.fused_cnt = 0;
    if (fused_cls < 0) return;
    bool first = true;
    for (int c = 0; c < bp.n_lane && c < BSW_MAX_LANE_CLASSES; ++c) {
        if (bp.lane_bits[c] != 8) continue;
        if (first) { pl.fused_off = pl.laneL_off[c]; first = false; }
        pl.fused_cnt += cl[c];
    }
}

/* See bsw_device.h (bsw_binparams.nsplit).  For chunks that do not fill the machine; BSW_NSPLIT = 0 / 1: never / every chunk with
 * 8-bit lane seeds (measurements, tests).  The host does not know which queries hold an N (it never reads a base): the lists keep
 * their counted sizes and the device leaves the places of the seeds it moved unfilled. */
static int nsplit_env()
{
    static const int env = getenv("BSW_NSPLIT") ? atoi(getenv("BSW_NSPLIT")) : -1;
    return env;
}
BSW_LOCAL bool nsplit_candidate(int kern, const bsw_binparams &bp, bool packed_ok, uint32_t n8, bool streaming)
{
    if (!bp.lane_on || !packed_ok || !n8 || nsplit_env() == 0) return false;
    if (nsplit_env() == 1) return true;
    return kern == BSW_KERNEL_AUTO && !streaming && n8 <= NSPLIT_MAX;
}
BSW_LOCAL bool nsplit_pays(int kern, const bsw_binparams &bp, bool packed_ok, uint32_t n8, bool streaming, double n_bases)
{
    if (!nsplit_candidate(kern, bp, packed_ok, n8, streaming)) return false;
    if (nsplit_env() == 1) return true;
    if (n_bases < 0) return n8 <= NSPLIT_MAX_BLIND;
    return n_bases <= (double)NLIST_WORK_MAX;
}

BSW_LOCAL void plan_nsplit(batch_plan &pl, bsw_binparams &bp, bool nsplit, uint32_t n_lane)
{
    pl.nsplit = nsplit ? 1 : 0;
    bp.nsplit = pl.nsplit;
    if (!nsplit) return;
    pl.nlist_off = pl.order_len;                         /* behind the redo list */
    pl.order_len += n_lane;
    pl.nlist_cnt_at = pl.order_len + 2 + BSW_MAX_WAVE_CLASSES;      /* (behind the words enqueue_parts zeroes) */
    bp.nlist_off = pl.nlist_off; bp.nlist_cap = n_lane; bp.nlist_cnt_at = pl.nlist_cnt_at;
    bp.fill_off = pl.lane_all_off; bp.fill_len = pl.order_len - pl.lane_all_off;      /* the lane lists, the redo list, the N list */
}

/* tasks[0..n) -> dt[0..n) (device task records), ro[0..n) (gather layout of the raw bytes), class counts -> plan */
/* One pass over the seeds of a chunk: validate, lay the 4-bit arena out, count the kernel classes.  `src(i, tmp, rc)`
 * hands out seed i as a bsw_task (a pointer into the caller's array, or `tmp` filled on the fly — bsw_submit_ref never
 * materialises its tasks); NULL = error rc.  rawoff gets the low 32 bits of every host pointer: when the chunk goes
 * out by direct DMA the pack kernel subtracts raw_bias, otherwise gather_offsets() replaces them. */
/* what one range of a chunk's seeds adds up to (the ranges of a chunk are walked side by side, then merged) */
struct range_acc {
    uint64_t words = 0, bytes = 0;                           /* sequence words / raw bytes the range needs */
    const uint8_t *lo = (const uint8_t *)UINTPTR_MAX, *hi = nullptr;
    uint32_t cw_all[BSW_MAX_WAVE_CLASSES] = {0}, cw[BSW_MAX_WAVE_CLASSES] = {0}, cw16[BSW_MAX_WAVE_CLASSES] = {0};
    uint32_t cl[BSW_MAX_LANE_CLASSES] = {0}, cr[BSW_MAX_LANE_CLASSES] = {0}, n_lane = 0, n16 = 0;
    uint8_t dep[BSW_MAX_LANE_CLASSES] = {0};
    uint64_t lane_work[BSW_MAX_LANE_CLASSES] = {0}, work8_l = 0, work8_r = 0;
    int h0_lo = INT_MAX, h0_hi = 0;
    /* packed input, first pass: the span of the words the range references and their number */
    const uint8_t *pl0 = (const uint8_t *)UINTPTR_MAX, *ph0 = nullptr;
    uint64_t psum = 0;
    int rc = 0;
    errs e;
};

template <class Src>
static int prepare_chunk_t(errs &e, const bsw_params *p, int kern, Src &&src, size_t n, bool dev_targets,
                           bsw_dtask *dt, bsw_rawoff *ro, chunk_info &ci, bool rev_left, bool packed = false, size_t idx0 = 0 /* index of the chunk's first seed in the caller's array: error texts name the caller's index */,
                           int threads = 1 /* ranges walked side by side (helper threads started here; their CPU time -> *helper_ns) */, std::atomic<uint64_t> *helper_ns = nullptr)
{
    ci.packed = packed;
    const int mx = mat_max(p->mat);
    int rc = fill_binparams(e, p, kern, ci.bp);
    if (rc) return rc;
    bsw_binparams &bp = ci.bp;
    /* class of a seed / a side by query length, looked up instead of searched per seed (the searches' data-dependent branches
     * were a fifth of this pass) — for chunks that repay the table, and for the lengths nearly every query has */
    constexpr int QTAB = 1024;
    const bool use_tab = n >= 4096;
    int8_t wcls[QTAB], lcls8[QTAB], lcls16[QTAB];
    if (use_tab)
        for (int q = 0; q < QTAB; ++q) {
            wcls[q] = (int8_t)bsw_wave_class_of(&bp, q);
            lcls8[q] = (int8_t)bsw_side_lane_class(&bp, 8, q);
            lcls16[q] = (int8_t)bsw_side_lane_class(&bp, 16, q);
        }
    const auto wave_cls = [&](int q) { return use_tab && q < QTAB ? (int)wcls[q] : bsw_wave_class_of(&bp, q); };
    const auto lane_cls = [&](int bits, int q) { return use_tab && q < QTAB ? (int)(bits == 8 ? lcls8[q] : lcls16[q]) : bsw_side_lane_class(&bp, bits, q); };
    /* THE PASS IS WALKED BY `threads` THREADS SIDE BY SIDE (round 6).  At 22 - 40 ns per seed it was what bounded every
     * PCIe-inclusive leg of bench.py: four slot threads = 100 - 140 M seeds/s of host-pass capacity against 240 M the GPU
     * takes (profiles/r6/e2e_host_pass.txt).  The seeds are cut into ranges; a first, cheap walk sizes every range (words,
     * bytes; packed input: the span of its words), a prefix sum gives every range its base, the second walk writes the records. */
    const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::min(std::max(threads, 1), BSW_REBASE_MAX), n / 32768 + 1));
    std::vector<range_acc> ra((size_t)T);
    const size_t per = (n + (size_t)T - 1) / (size_t)T;
    auto run_ranges = [&](auto &&fn) {
        std::vector<std::thread> th;
        for (int k = 1; k < T; ++k)
            th.emplace_back([&, k]() { const uint64_t c0 = thread_cpu_ns(); fn(k); if (helper_ns) *helper_ns += thread_cpu_ns() - c0; });
        fn(0);
        for (auto &t : th) t.join();
    };
    /* ONE walk.  Word offsets are written RELATIVE TO THE RANGE (every range counts from 0) and made absolute on the device:
     * bsw_rebase_kernel adds the range's base — known only when all ranges are done — to the four offsets of each of its seeds
     * (44 bytes read-modify-write per seed at HBM speed: nothing beside a second host walk, which is what the first version of
     * the parallel pass spent half its time on).  Packed input that can be DMA'd as it lies takes its offsets from the POINTERS:
     * rawoff (unused by packed input otherwise) carries (pointer >> 3) mod 2^32 per sequence, and the device adds
     * -(arena base >> 3): whether the arena is compact and registered is decided after the walk, from the span it found. */
    /* ---- walk 2: validate, lay out, count ---- */
    run_ranges([&](int k) {
        range_acc &A = ra[(size_t)k];
        errs &el = A.e;
        bsw_task tmp;
        int rcl = 0;
        const size_t i0 = std::min(n, per * (size_t)k), i1 = std::min(n, i0 + per);
        uint64_t acc = 0, accb = 0;
        const uint8_t *lo = (const uint8_t *)UINTPTR_MAX, *hi = nullptr;
        /* H5/H6 per query length, computed once per length and range (two integer divisions each: per seed they were most of
         * this pass) */
        std::vector<uint16_t> gl5((size_t)BSW_MAX_QLEN + 1, 0), gl3((size_t)BSW_MAX_QLEN + 1, 0);
        const auto glim = [&](std::vector<uint16_t> &tab, int qlen, int clip) -> uint16_t {
            uint16_t &v = tab[(size_t)qlen];
            if (!v) v = (uint16_t)gap_limit(p, mx, qlen, clip);       /* >= 1: zero means not computed yet */
            return v;
        };
        auto span = [&](const uint8_t *sq, int len) {
            if (len > 0) { if (sq < lo) lo = sq; if (sq + len > hi) hi = sq + len; }
        };
        auto bad = [&](int code) { A.rc = code; };
    for (size_t i = i0; i < i1; ++i) {
        const bsw_task *tp = src(i, tmp, rcl);
        if (!tp) { A.rc = rcl; return; }
        const bsw_task &t = *tp;
        if (t.lqlen < 0 || t.rqlen < 0 || t.ltlen < 0 || t.rtlen < 0)
            return bad(fail(el, BSW_E_INVAL, "task %zu: negative length", i + idx0));
        if (t.lqlen > BSW_MAX_QLEN || t.rqlen > BSW_MAX_QLEN || t.ltlen > BSW_MAX_TLEN || t.rtlen > BSW_MAX_TLEN)
            return bad(fail(el, BSW_E_LIMIT, "task %zu: length beyond BSW_MAX_QLEN/BSW_MAX_TLEN", i + idx0));
        if (t.h0 <= 0) return bad(fail(el, BSW_E_INVAL, "task %zu: h0 must be > 0", i + idx0));
        if ((int64_t)t.h0 + (int64_t)(t.lqlen + t.rqlen) * mx >= BSW_MAX_SCORE)
            return bad(fail(el, BSW_E_LIMIT, "task %zu: score range beyond BSW_MAX_SCORE", i + idx0));
        if ((t.lqlen && (!t.lquery || (t.ltlen && !t.ltarget && !dev_targets))) ||
            (t.rqlen && (!t.rquery || (t.rtlen && !t.rtarget && !dev_targets))))
            return bad(fail(el, BSW_E_INVAL, "task %zu: NULL sequence pointer", i + idx0));
        if (t.wlim_l < 0 || t.wlim_r < 0) return bad(fail(el, BSW_E_INVAL, "task %zu: negative wlim", i + idx0));
        bsw_dtask d;                                /* built here, stored once: dt[] is write-combined staging */
        bsw_rawoff r;
        memset(&d, 0, sizeof(d));
        memset(&r, 0, sizeof(r));
        if (packed) {
            /* 16 bases per uint64 already (base k in bits [4k, 4k+3], codes 0-3 = ACGT, 4-7 = N), every sequence on an
             * 8-byte boundary: the spans are whole words, rawoff keeps the pointers' low bits until the arena base is known */
            if ((t.lqlen && (((uintptr_t)t.lquery | (t.ltlen ? (uintptr_t)t.ltarget : 0)) & 7)) ||
                (t.rqlen && (((uintptr_t)t.rquery | (t.rtlen ? (uintptr_t)t.rtarget : 0)) & 7)))
                return bad(fail(el, BSW_E_INVAL, "task %zu: packed sequences must start on 8-byte boundaries", i + idx0));
            /* d: offsets for the gathered layout (range-relative); r: (pointer >> 3) mod 2^32, what the device turns into
             * offsets into the arena when it is DMA'd as it lies (an empty target takes its query's word: word 0 of a target
             * may be read even when no row is) */
            if (t.lqlen) {
                d.lq_off = (uint32_t)acc; acc += nwords(t.lqlen);
                d.lt_off = (uint32_t)acc; acc += nwords(t.ltlen);
                r.lq = (uint32_t)((uintptr_t)t.lquery >> 3);
                r.lt = (uint32_t)((uintptr_t)(t.ltlen ? t.ltarget : t.lquery) >> 3);
                accb += 8ull * (nwords(t.lqlen) + nwords(t.ltlen));
                span(t.lquery, 8 * (int)nwords(t.lqlen)); span(t.ltarget, 8 * (int)nwords(t.ltlen));
            }
            if (t.rqlen) {
                d.rq_off = (uint32_t)acc; acc += nwords(t.rqlen);
                d.rt_off = (uint32_t)acc; acc += nwords(t.rtlen);
                r.rq = (uint32_t)((uintptr_t)t.rquery >> 3);
                r.rt = (uint32_t)((uintptr_t)(t.rtlen ? t.rtarget : t.rquery) >> 3);
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
        ro[i] = r;
        /* class counts (the device sorts with the same functions) */
        const int qm = t.lqlen > t.rqlen ? t.lqlen : t.rqlen;
        const int wc = wave_cls(qm);
        if (wc < 0) return bad(fail(el, BSW_E_LIMIT, "task %zu: no kernel class", i + idx0));
        ++A.cw_all[wc];
        const int bits = bsw_seed_lane_bits(&bp, t.lqlen, t.rqlen, t.h0);
        if (!bits) ++A.cw[wc];
        else {
            ++A.n_lane;
            if (bits == 16) { ++A.n16; ++A.cw16[wc]; }
            int lc = -1;
            if (t.lqlen) {
                const int c = lc = lane_cls(bits, t.lqlen);
                if (c < 0) return bad(fail(el, BSW_E_LIMIT, "task %zu: no lane class", i + idx0));
                ++A.cl[c];
                A.lane_work[c] += (uint64_t)t.lqlen;
                if (bits == 8) A.work8_l += (uint64_t)t.lqlen;
                if (bits == 8) { A.h0_lo = std::min(A.h0_lo, t.h0); A.h0_hi = std::max(A.h0_hi, t.h0); }      /* (the 8-bit seeds': a few 16-bit ones — h0 in the hundreds — would widen the buckets to nothing) */
            }
            if (t.rqlen) {
                const int c = lane_cls(bits, t.rqlen);
                if (c < 0) return bad(fail(el, BSW_E_LIMIT, "task %zu: no lane class", i + idx0));
                ++A.cr[c];
                A.lane_work[c] += (uint64_t)t.rqlen;
                if (bits == 8) A.work8_r += (uint64_t)t.rqlen;
                if (lc >= 0) A.dep[lc] |= (uint8_t)(1u << c);
            }
        }
    }
        A.lo = lo; A.hi = hi;
        A.words = acc; A.bytes = accb;
    });
    /* merge the ranges (the first failure in seed order is the one reported) */
    for (const range_acc &A : ra) if (A.rc) { if (!A.e.msg.empty()) e = A.e; return A.rc; }
    uint64_t acc = 0, accb = 0;
    const uint8_t *lo = (const uint8_t *)UINTPTR_MAX, *hi = nullptr;
    uint32_t cw_all[BSW_MAX_WAVE_CLASSES] = {0}, cw[BSW_MAX_WAVE_CLASSES] = {0}, cw16[BSW_MAX_WAVE_CLASSES] = {0};
    uint32_t cl[BSW_MAX_LANE_CLASSES] = {0}, cr[BSW_MAX_LANE_CLASSES] = {0}, n_lane = 0, n16 = 0;
    uint8_t dep[BSW_MAX_LANE_CLASSES] = {0};
    uint64_t lane_work[BSW_MAX_LANE_CLASSES] = {0};          /* sum of query lengths per lane class (narrow_fold) */
    uint64_t work8_l = 0, work8_r = 0;                       /* ... of the 8-bit left / right sides (decide_lane_mode) */
    int h0_lo = INT_MAX, h0_hi = 0;                          /* h0 range of the left sides that go to lane classes (bsw_h0_bucket) */
    for (const range_acc &A : ra) {
        acc += A.words; accb += A.bytes;
        if (A.hi) { if (A.lo < lo) lo = A.lo; if (A.hi > hi) hi = A.hi; }
        for (int c = 0; c < BSW_MAX_WAVE_CLASSES; ++c) { cw_all[c] += A.cw_all[c]; cw[c] += A.cw[c]; cw16[c] += A.cw16[c]; }
        for (int c = 0; c < BSW_MAX_LANE_CLASSES; ++c) { cl[c] += A.cl[c]; cr[c] += A.cr[c]; dep[c] |= A.dep[c]; lane_work[c] += A.lane_work[c]; }
        n_lane += A.n_lane; n16 += A.n16; work8_l += A.work8_l; work8_r += A.work8_r;
        h0_lo = std::min(h0_lo, A.h0_lo); h0_hi = std::max(h0_hi, A.h0_hi);
    }
    if (acc >= (1ull << 32)) return fail(e, BSW_E_LIMIT, "batch sequence arena beyond 2^32 words; split the batch");
    if (accb >= (1ull << 32) - RAW_SLACK) return fail(e, BSW_E_LIMIT, "batch holds more than 4 GiB of bases; split the batch");
    bool group = false, packed_ok = false;
    {
        bsw_dparams dpx;
        errs quiet;
        packed_ok = check_params(quiet, p, &dpx) == BSW_OK && bsw::lane_class_finishes(0, dpx, p->variant);
        group = decide_lane_mode(kern, packed_ok, bp, n_lane, n16, cl, cr, cw, cw16, dep, work8_l, work8_r, ci.streaming);
    }
    if (bp.lane_on && narrow_foldable(bp)) {
        uint64_t all8 = 0;
        for (int c = 0; c < bp.n_lane; ++c) if (bp.lane_bits[c] == bp.lane_bits[0]) all8 += lane_work[c];
        /* (the group kernel runs one instantiation for both 8-bit classes up to 136 columns: one launch per side) */
        if (group || (double)lane_work[0] < narrow_min_share() * (double)all8) narrow_fold(bp, cl, cr, dep);
    }
    if (!bp.lane_on) {
        memcpy(cw, cw_all, sizeof(cw));
        memset(cl, 0, sizeof(cl));
        memset(cr, 0, sizeof(cr));
        n_lane = 0;
    }
    const int fused_cls = fuse_lists(bp, kern, group, packed_ok, group ? n_lane : n_lane - n16, cl, cr, ci.streaming);
    if (bp.lane_on && h0_hi >= h0_lo) bsw_set_h0_buckets(&bp, h0_lo, h0_hi);
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
    plan_fused(pl, bp, fused_cls, group, cl);
    {
        /* the share of seeds with an N in a query: the pass never reads a base, a SAMPLE of the chunk's seeds does (both queries of
         * up to 512 seeds, evenly spaced; ~0.1 ms) — only for chunks the split could pay for.  x the chunk's query bases = the work
         * the general kernel would get */
        const uint32_t n8 = group ? n_lane : n_lane - n16;
        double nfrac = -1.0;
        if (nsplit_candidate(kern, bp, packed_ok, n8, ci.streaming) && nsplit_env() != 1) {
            const auto has_n = [&](const uint8_t *q, int len, bool backwards) -> bool {
                if (!q || len <= 0) return false;
                if (packed) {
                    const uint64_t *w = (const uint64_t *)q;
                    const int nw = (len + 15) >> 4;
                    uint64_t acc = 0;
                    for (int k = 0; k < nw; ++k) {
                        uint64_t v = w[k];
                        if (k == nw - 1 && (len & 15)) v &= (1ull << (4 * (len & 15))) - 1ull;
                        acc |= v;
                    }
                    return (acc & 0x4444444444444444ull) != 0;
                }
                const uint8_t *b = backwards ? q - (len - 1) : q;
                for (int k = 0; k < len; ++k) if (b[k] >= 4) return true;
                return false;
            };
            const size_t S = std::min<size_t>(n, 512);
            size_t hits = 0, seen = 0;
            bsw_task tmp;
            int rcl = 0;
            for (size_t k = 0; k < S; ++k) {
                const bsw_task *tp = src(k * n / S, tmp, rcl);
                if (!tp) continue;
                ++seen;
                hits += (has_n(tp->lquery, tp->lqlen, rev_left) || has_n(tp->rquery, tp->rqlen, false)) ? 1 : 0;
            }
            nfrac = (seen ? (double)hits / (double)seen : 0.0) * (double)(work8_l + work8_r);      /* -> the list's query bases */
        }
        plan_nsplit(pl, bp, nsplit_pays(kern, bp, packed_ok, n8, ci.streaming, nfrac), n_lane);
    }
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
        /* (span and bytes of packed input are whole words: lo / hi / sum_len were taken over 8 * nwords) */
        ci.direct = spanb > 0 && spanb < (1ull << 32) - RAW_SLACK && spanb <= 2 * ci.sum_len + (1u << 20) && is_registered(lo, spanb);
        ci.rev_left = false;
        ci.raw_bias = 0;
        if (ci.direct) ci.words = spanb >> 3;
    }
    /* what bsw_rebase_kernel has to do to the records on the device */
    ci.rb = bsw_rebase();
    ci.rb.per = (uint32_t)std::max<size_t>(per, 1);
    ci.rb.nr = (uint32_t)T;
    uint64_t run = 0;
    for (int k = 0; k < T; ++k) { ci.rb.base[k] = (uint32_t)run; run += ra[(size_t)k].words; }
    ci.rb.use_ro = packed && ci.direct ? 1u : 0u;
    ci.rb.delta = ci.rb.use_ro ? (uint32_t)(0u - (uint32_t)((uintptr_t)lo >> 3)) : 0u;
    ci.rb_on = T > 1 || ci.rb.use_ro;
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
                         bsw_dtask *dt, bsw_rawoff *ro, chunk_info &ci, bool rev_left = false, bool packed = false, size_t idx0 = 0,
                         int threads = 1, std::atomic<uint64_t> *helper_ns = nullptr)
{
    int rc = prepare_chunk_t(e, p, kern, [tasks](size_t i, bsw_task &, int &) { return tasks + i; }, n, dev_targets, dt, ro, ci, rev_left, packed, idx0, threads, helper_ns);
    if (!rc && !ci.direct && !packed) gather_offsets(tasks, n, dev_targets, ro);
    return rc;
}

/* packed sequences that are not in registered memory: their words go to the pinned staging arena in seq layout */
static void gather_packed(const bsw_task *tasks, const bsw_dtask *dt, size_t n, uint64_t *dst, const bsw_rebase &rb)
{
    for (size_t i = 0; i < n; ++i) {
        const bsw_task &t = tasks[i];
        bsw_dtask d = dt[i];
        const uint32_t base = rb.base[std::min<size_t>(i / rb.per, rb.nr - 1)];       /* (the records hold range-relative offsets until the device rebases them) */
        d.lq_off += base; d.lt_off += base; d.rq_off += base; d.rt_off += base;
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
/* helper_ns: CPU time of the helper threads started here (the calling thread accounts for itself) */
static void gather_raw(const bsw_task *tasks, const bsw_rawoff *ro, size_t n, bool dev_targets, uint8_t *dst, int threads, bool rev_left = false,
                       std::atomic<uint64_t> *helper_ns = nullptr)
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
        if (lo < hi) th.emplace_back([&work, lo, hi, helper_ns]() { const uint64_t c0 = thread_cpu_ns(); work(lo, hi); if (helper_ns) *helper_ns += thread_cpu_ns() - c0; });
    }
    work(0, std::min(n, per));
    for (auto &t : th) t.join();
}

/* ---- device side of a chunk: DMA, pack, (fetch), bin, and optionally the DP kernels ---------- */
/* pipeline: the caller is a slot of the streaming pipeline — its chunks' tails are filled by the other slots' chunks, and a
 * stream that waits for a flag holds up whatever shares its hardware queue, so the tail-fill mode stays out of it */
BSW_LOCAL const fork_t *fork_for(const bsw_ctx *ctx, hipStream_t s, bool pipeline)
{
    for (const dev_state &d : ctx->devs)
        for (size_t k = 0; k < d.streams.size(); ++k)
            if (d.streams[k] == s) return k < d.forks.size() && d.forks[k].ok && !(pipeline && d.forks[k].mode == 2) ? &d.forks[k] : nullptr;
    return nullptr;
}

/* one batch (or one part of a split resident batch) as enqueue_parts() sees it */
struct lane_job {
    const bsw_dparams *P;
    int variant;
    const uint64_t *d_seq;
    const bsw_dtask *d_tasks;
    uint32_t *d_order;
    const batch_plan *pl;
    bsw_result *d_out;
    bsw_pair *d_pair;
    uint64_t *launches;
};

/* The general-kernel launches, the lane launches and the pair decision of `nj` independent parts on stream s.
 *
 * THE LANE LAUNCHES AS A CHAIN (fork mode 2).  Per part: left sides narrowest class first, then right sides
 * widest first — a long right side belongs to a short left side and the other way round, so every launch's seeds are ready
 * as early as they can be; the left launches of all parts before the right ones.  The links go round the device's slot streams (up to four; the others are idle while stream 0 runs a resident batch or a synchronous chunk).  A link
 * waits for the events of the left-side launches of its part that hold its seeds (plan.dep) and for the FLAG of the link
 * before it: the two-seeds-per-lane kernels count their started workgroups in a device word, a sleeping wave in front of the follower polls it, and once the count
 * reaches the grid size every slot that frees up stays free (another kernel's flag is raised behind it) — the follower takes
 * the ragged end of its predecessor and never a slot one of the predecessor's waves could have had.  Worth its stream
 * operations only where some link does NOT depend on the link before it (otherwise it would have to wait for the end anyway):
 * 250 bp reads (136- and 232-column classes: 16.6 - 17.1 ms per 1 M seeds -> 16.0, gpurun_out/r7c).  Several parts: a resident
 * batch uploaded as two halves so that the right sides of one run in the end of the other's left sides was measured and
 * LOSES (mixed PE bins 4.42 -> 4.91 ms: every launch has the floor of its longest waves, 1.39 ms, and a half-batch launch is
 * little more than that floor; longest-first over the whole batch packs better than any chain of parts) — the interface
 * stays general, one part is what runs. */
static int enqueue_parts(errs &e, const lane_job *jobs, int nj, hipStream_t s, const fork_t *fk)
{
    const int nc = bsw::wave_class_count(), nlc = bsw::lane_class_count();
    bool nlist_side = false;                                       /* the N list's launch (bsw_binparams.nsplit) runs on a borrowed stream */
    for (int j = 0; j < nj; ++j) {
        const lane_job &J = jobs[j];
        const batch_plan &pl = *J.pl;
        /* device words behind the order lists: [0] the redo list's length, [1 + c] the work counter of wave class c's launch,
         * [1 + BSW_MAX_WAVE_CLASSES] the redo launch's — zeroed here, on the stream, before anything counts in them */
        uint32_t *redo_cnt = J.d_order + pl.order_len;
        HIPCHK(e, hipMemsetAsync(redo_cnt, 0, (2 + BSW_MAX_WAVE_CLASSES) * sizeof(uint32_t), s));      /* (the word behind them: the N list's length, plan.nlist_cnt_at) */
        /* The general-kernel classes of a part side by side (round 5).  A batch below the lane kernels' threshold is a few
         * thousand wavefronts in two or three classes (PE seeds: 64 / 128 / 192 columns); launched one after the other each class
         * leaves most of the machine idle for one wave's lifetime.  With idle streams at hand (the fork set of stream 0: resident
         * batches, synchronous chunks) class k goes to stream k, the slot stream joins them: 13 k PE mixed seeds 0.77 -> 0.55 ms
         * (profiles/r5/crossover_mixed_bins_general_forked.json).  The reference's task_parse keeps its 20 PEs busy from any batch
         * (sw_pe_array_task_parse.v:1600-1648). */
        int nwave = 0;
        for (int c = 0; c < nc; ++c) nwave += pl.wave_start[c + 1] - pl.wave_start[c] ? 1 : 0;
        static const bool no_wfork = getenv("BSW_NO_WAVE_FORK") != nullptr;        /* (measurements) */
        const bool wfork = !no_wfork && nwave >= 2 && fk && fk->ok && fk->naux > 0 && (fk->mode == 1 || fk->mode == 2);
        bool used[BSW_FORK_AUX] = {false};
        if (wfork) {
            HIPCHK(e, hipEventRecord(fk->ev_fork, s));                  /* the input DMAs, pack, bins and the zeroed counters */
            for (int a = 0; a < fk->naux; ++a) HIPCHK(e, hipStreamWaitEvent(fk->aux[a], fk->ev_fork, 0));
        }
        int k = 0;
        for (int c = nc - 1; c >= 0; --c) {                             /* widest class first: its waves run longest */
            const uint32_t cnt = pl.wave_start[c + 1] - pl.wave_start[c];
            if (!cnt) continue;
            hipStream_t ws = s;
            if (wfork && k > 0) { const int a = (k - 1) % fk->naux; ws = fk->aux[a]; used[a] = true; }
            ++k;
            HIPCHK(e, bsw::launch_wave(c, J.variant, *J.P, J.d_seq, J.d_tasks, J.d_order + pl.wave_start[c], cnt, nullptr, redo_cnt + 1 + c, J.d_out, ws));
            if (J.d_pair) HIPCHK(e, bsw::launch_pairs_from_results(J.d_order + pl.wave_start[c], cnt, nullptr, J.d_out, J.d_pair, ws));
            if (J.launches) ++*J.launches;
        }
        if (wfork)
            for (int a = 0; a < fk->naux; ++a)
                if (used[a]) {
                    HIPCHK(e, hipEventRecord(fk->ev_left[a], fk->aux[a]));
                    HIPCHK(e, hipStreamWaitEvent(s, fk->ev_left[a], 0));
                }
        if (pl.nsplit && pl.lane_all_cnt) {
            /* bsw_binparams.nsplit: the lane seeds with an N in a query, one wavefront each, BESIDE the lane launches — on a borrowed
             * stream, joined behind them (below); without one, in front of them on the slot stream */
            const bool side = nj == 1 && fk && fk->ok && fk->naux > 0 && (fk->mode == 1 || fk->mode == 2);
            hipStream_t ns = s;
            if (side) {
                /* (NOT the last borrowed stream: it shares a hardware queue with the slot stream — the list's launch then
                 * ran in front of the lane launches instead of beside them, 16 k PE seeds 0.86 ms against 0.70) */
                ns = fk->aux[fk->naux > 1 ? 1 : 0];                   /* (the second: a launch chain's first follower takes the first) */
                HIPCHK(e, hipEventRecord(fk->ev_fork_r, s));            /* the bins, the zeroed counters */
                HIPCHK(e, hipStreamWaitEvent(ns, fk->ev_fork_r, 0));
            }
            HIPCHK(e, bsw::launch_wave(pl.redo_cls, J.variant, *J.P, J.d_seq, J.d_tasks, J.d_order + pl.nlist_off, pl.lane_all_cnt, J.d_order + pl.nlist_cnt_at, nullptr, J.d_out, ns));
            if (J.d_pair) HIPCHK(e, bsw::launch_pairs_from_results(J.d_order + pl.nlist_off, pl.lane_all_cnt, J.d_order + pl.nlist_cnt_at, J.d_out, J.d_pair, ns));
            if (side) HIPCHK(e, hipEventRecord(fk->ev_nlist, ns));
            nlist_side = side;
            if (J.launches) ++*J.launches;
        }
    }
    /* Does every lane launch of a part run a kernel that finishes its seeds itself (bsw_fin: the launch that computes a seed's
     * last side takes the pair-level decision)?  Then bsw_pair_finalize is not launched for it: one launch and a 96-byte
     * read-modify-write per seed less per step.  One class on round 1's kernel (16-bit rows, odd N scores) keeps the launch. */
    constexpr int MAXJ = 4;
    bsw_fin fins[MAXJ];
    for (int j = 0; j < nj && j < MAXJ; ++j) {
        const lane_job &J = jobs[j];
        const batch_plan &pl = *J.pl;
        bool all = nj <= MAXJ;
        for (int c = 0; c < nlc && all; ++c)
            if ((pl.laneL_off[c + 1] - pl.laneL_off[c] || pl.laneR_off[c + 1] - pl.laneR_off[c]) && !bsw::lane_class_finishes(c, *J.P, J.variant)) all = false;
        fins[j].redo = J.d_order + pl.redo_off; fins[j].redo_cnt = J.d_order + pl.order_len; fins[j].pairs = J.d_pair; fins[j].on = all ? 1 : 0;
        fins[j].group = pl.lane_group;
    }
    auto fin_of = [&](int j) -> const bsw_fin * { return j < MAXJ ? &fins[j] : nullptr; };
    struct link { int job, side, cls; uint32_t off, cnt; };
    constexpr int MAXL = 2 * BSW_MAX_LANE_CLASSES;
    link chain[MAXL];
    int nchain = 0;
    bool chain_pays = false;
    bool any_group = false;
    for (int j = 0; j < nj; ++j) any_group = any_group || jobs[j].pl->lane_group || jobs[j].pl->fused_cls >= 0;      /* (nor for a fused launch) */
    if (fk && fk->mode == 2 && !any_group) {        /* (the group kernel's launches are short: nothing for a chain to fill) */
        bool fits = true;
        for (int side = 0; side < 2; ++side)
            for (int j = 0; j < nj; ++j) {
                const batch_plan &pl = *jobs[j].pl;
                if (!pl.lane_all_cnt) continue;
                for (int k = 0; k < nlc; ++k) {
                    const int c = side ? nlc - 1 - k : k;
                    const uint32_t *off = side ? pl.laneR_off : pl.laneL_off;
                    if (off[c + 1] - off[c] == 0) continue;
                    if (nchain == MAXL) { fits = false; break; }
                    chain[nchain++] = link{j, side, c, off[c], off[c + 1] - off[c]};
                }
            }
        auto depends = [&](const link &x, const link &on) {     /* x cannot start before `on` is done */
            return x.job == on.job && x.side == 1 && on.side == 0 && ((jobs[x.job].pl->dep[on.cls] >> x.cls) & 1);
        };
        for (int i = 0; fits && i + 1 < nchain; ++i)
            chain_pays = chain_pays || (bsw::lane_class_signals_tail(chain[i].cls) && !depends(chain[i + 1], chain[i]));
    }
    if (chain_pays) {
        const hipStream_t rot[4] = {s, fk->aux[0], fk->aux[1], fk->aux[2]};
        const int nrot = 1 + fk->naux;                               /* 2 .. 4 */
        hipStream_t on[MAXL];                                        /* the stream of link i */
        uint32_t target = 1u;                                        /* what the previous link's flag reaches */
        HIPCHK(e, hipMemsetAsync(fk->flag_mem, 0, (size_t)MAXL * 64 * sizeof(uint32_t), s));
        HIPCHK(e, hipEventRecord(fk->ev_fork, s));                  /* everything queued on s so far (input DMAs, pack, bins, general kernels), and the flags are down */
        for (int a = 0; a < fk->naux; ++a) HIPCHK(e, hipStreamWaitEvent(fk->aux[a], fk->ev_fork, 0));
        for (int i = 0; i < nchain; ++i) {
            const link &k = chain[i];
            const lane_job &J = jobs[k.job];
            const hipStream_t ks = on[i] = rot[i % nrot];
            for (int h = 0; h < i; ++h) {
                const link &l = chain[h];
                if (on[h] != ks && l.job == k.job && l.side == 0 && k.side == 1 && ((J.pl->dep[l.cls] >> k.cls) & 1)) HIPCHK(e, hipStreamWaitEvent(ks, fk->ev_link[h], 0));
            }
            /* BSW_CHAIN_SELFTEST (tests only): a target no count reaches, so that EVERY wait runs into its deadline — what a tool
             * that runs the waiter before the launch raising its word does; results must not change, bsw_chain_timeouts counts */
            static const bool selftest = getenv("BSW_CHAIN_SELFTEST") != nullptr;
            if (i > 0) HIPCHK(e, bsw::launch_wait_count(fk->flag(i - 1), selftest ? 0xffffffffu : target, fk->expired(), ks));
            HIPCHK(e, bsw::launch_lane(k.cls, J.variant, *J.P, k.side, J.d_seq, J.d_tasks, J.d_order + k.off, k.cnt, J.d_out, ks, i + 1 < nchain ? fk->flag(i) : nullptr, &target, fin_of(k.job)));
            HIPCHK(e, hipEventRecord(fk->ev_link[i], ks));
            if (J.launches) ++*J.launches;
        }
        for (int i = 0; i < nchain; ++i)                              /* join: the slot stream waits for whatever ran elsewhere */
            if (on[i] != s) HIPCHK(e, hipStreamWaitEvent(s, fk->ev_link[i], 0));
    }
    for (int j = 0; j < nj; ++j) {
        const lane_job &J = jobs[j];
        const batch_plan &pl = *J.pl;
        if (!pl.lane_all_cnt) continue;
        const bsw_dparams &P = *J.P;
        const int variant = J.variant;
        const uint64_t *d_seq = J.d_seq;
        const bsw_dtask *d_tasks = J.d_tasks;
        uint32_t *d_order = J.d_order, *redo_cnt = J.d_order + pl.order_len;
        bsw_result *d_out = J.d_out;
        bsw_pair *d_pair = J.d_pair;
        uint64_t *launches = J.launches;
        if (pl.fused_cls >= 0 && pl.fused_cnt) {                       /* both sides of every 8-bit seed in one launch over their left lists */
            HIPCHK(e, bsw::launch_lane(pl.fused_cls, variant, P, 2, d_seq, d_tasks, d_order + pl.fused_off, pl.fused_cnt, d_out, s, nullptr, nullptr, fin_of(j)));
            if (launches) ++*launches;
        }
        if (!chain_pays) {
        const fork_t *fk1 = fk && fk->mode == 1 ? fk : nullptr;       /* (mode 2 without a chain that pays: plain launches on s) */
        /* The classes of a side side by side: the k-th non-empty class of a side (widest first: its waves run longest) goes
         * to stream k — the slot stream, then the auxiliary ones.  A right-side launch waits for exactly the left-side
         * launches that hold one of its seeds (plan.dep); streams are in-order, so only other streams' launches need an event. */
        hipStream_t lstream[BSW_MAX_LANE_CLASSES] = {nullptr}, rstream[BSW_MAX_LANE_CLASSES] = {nullptr};
        int nl = 0, nr = 0;
        const auto in_fused = [&](int c) { return pl.fused_cls >= 0 && bsw::lane_class_bits(c) == 8; };      /* (their seeds ran above) */
        for (int c = nlc - 1; c >= 0; --c) {
            if (in_fused(c)) continue;
            if (pl.laneL_off[c + 1] - pl.laneL_off[c]) { lstream[c] = (fk1 && nl > 0 && nl <= BSW_FORK_AUX) ? fk1->aux[nl - 1] : s; ++nl; }
            if (pl.laneR_off[c + 1] - pl.laneR_off[c]) { rstream[c] = (fk1 && nr > 0 && nr <= BSW_FORK_AUX) ? fk1->aux[nr - 1] : s; ++nr; }
        }
        const bool forked = fk1 && (nl > 1 || nr > 1);
        if (forked) {
            HIPCHK(e, hipEventRecord(fk1->ev_fork, s));                 /* everything queued on s so far (input DMAs, pack, bins) */
            for (int a = 0; a < BSW_FORK_AUX; ++a) HIPCHK(e, hipStreamWaitEvent(fk1->aux[a], fk1->ev_fork, 0));
        }
        for (int c = nlc - 1; c >= 0; --c) {
            const uint32_t cnt = pl.laneL_off[c + 1] - pl.laneL_off[c];
            if (!cnt || in_fused(c)) continue;
            HIPCHK(e, bsw::launch_lane(c, variant, P, 0, d_seq, d_tasks, d_order + pl.laneL_off[c], cnt, d_out, lstream[c], nullptr, nullptr, fin_of(j)));
            if (forked) HIPCHK(e, hipEventRecord(fk1->ev_left[c], lstream[c]));
            if (launches) ++*launches;
        }
        for (int c = nlc - 1; c >= 0; --c) {
            const uint32_t cnt = pl.laneR_off[c + 1] - pl.laneR_off[c];
            if (!cnt || in_fused(c)) continue;
            if (forked)
                for (int lc = 0; lc < nlc; ++lc)
                    if (lstream[lc] && lstream[lc] != rstream[c] && ((pl.dep[lc] >> c) & 1)) HIPCHK(e, hipStreamWaitEvent(rstream[c], fk1->ev_left[lc], 0));
            HIPCHK(e, bsw::launch_lane(c, variant, P, 1, d_seq, d_tasks, d_order + pl.laneR_off[c], cnt, d_out, rstream[c], nullptr, nullptr, fin_of(j)));
            if (forked && rstream[c] != s) HIPCHK(e, hipEventRecord(fk1->ev_right[c], rstream[c]));
            if (launches) ++*launches;
        }
        if (forked) {                                                   /* join: the slot stream waits for whatever ran elsewhere */
            for (int c = 0; c < nlc; ++c) {
                if (lstream[c] && lstream[c] != s) HIPCHK(e, hipStreamWaitEvent(s, fk1->ev_left[c], 0));
                if (rstream[c] && rstream[c] != s) HIPCHK(e, hipStreamWaitEvent(s, fk1->ev_right[c], 0));
            }
        }
        }
        const bool folded = fin_of(j) && fin_of(j)->on;
        if (!folded)
            HIPCHK(e, bsw::launch_finalize(P, d_tasks, d_order + pl.lane_all_off, pl.lane_all_cnt, d_out,
                                           d_order + pl.redo_off, redo_cnt, d_pair, s));
        /* seeds whose first band try was not final: recompute from scratch, one wavefront each */
        HIPCHK(e, bsw::launch_wave(pl.redo_cls, variant, P, d_seq, d_tasks, d_order + pl.redo_off, pl.lane_all_cnt,
                                   redo_cnt, redo_cnt + 1 + BSW_MAX_WAVE_CLASSES, d_out, s));
        if (d_pair) HIPCHK(e, bsw::launch_pairs_from_results(d_order + pl.redo_off, pl.lane_all_cnt, redo_cnt, d_out, d_pair, s));
        if (launches) *launches += folded ? 1 : 2;
    }
    if (nlist_side) HIPCHK(e, hipStreamWaitEvent(s, fk->ev_nlist, 0));       /* join: the N list's launch */
    return BSW_OK;
}

BSW_LOCAL int enqueue_batch(errs &e, const bsw_dparams &P, int variant, const uint64_t *d_seq, const bsw_dtask *d_tasks,
                         uint32_t *d_order, const batch_plan &pl, bsw_result *d_out, hipStream_t s, uint64_t *launches,
                         const fork_t *fk, bsw_pair *d_pair)
{
    const lane_job j = {&P, variant, d_seq, d_tasks, d_order, &pl, d_out, d_pair, launches};
    return enqueue_parts(e, &j, 1, s, fk);
}

/* the raw bytes and the task records are in st.h_* (or the caller's registered arena): move them, pack, bin */


static int stage_device(errs &e, stage_t &st, hipStream_t s, const chunk_info &ci, size_t n, bool dev_targets,
                        const bsw_ref *ref, uint64_t *h2d_bytes, const gate_turn *turn = nullptr, size_t dev_index = 0)
{
    const size_t n_desc = ref ? n : 0;              /* st.h_desc: one bsw_refx per seed */
    const size_t rawb = ci.packed ? 0 : (ci.direct ? (size_t)(ci.hi - ci.lo) : ci.sum_len);
    hipError_t he;
    if ((he = st.d_raw.reserve(rawb + RAW_FRONT + RAW_SLACK)) != hipSuccess || (he = st.d_seq.reserve(ci.words + 4)) != hipSuccess ||
        (he = st.d_tasks.reserve(n + 1)) != hipSuccess || (he = st.d_roff.reserve(n + 1)) != hipSuccess ||
        (he = st.d_order.reserve(order_capacity(n))) != hipSuccess || (he = st.d_bins.reserve(BSW_BIN_WORDS)) != hipSuccess ||
        (he = st.d_nflag.reserve(n + 4)) != hipSuccess || (he = st.d_keys.reserve(n + 1)) != hipSuccess ||
        (he = st.d_out.reserve(n + 1)) != hipSuccess || (n_desc && (he = st.d_desc.reserve(n_desc)) != hipSuccess))
        return fail(e, BSW_E_NOMEM, "device staging: %s", hipGetErrorString(he));
    {
        std::unique_lock<std::mutex> lk;
        if (turn) {
            lk = std::unique_lock<std::mutex>(turn->gate->mu);
            /* the turns of a device are taken strictly in order, also by chunks that have nothing to do any more (their submit
             * failed elsewhere): chunks of OTHER submits queue behind them */
            turn->gate->cv.wait(lk, [&]() { return turn->gate->next == turn->seq; });
            hipError_t we = hipSuccess;
            if (!*turn->abort_flag && turn->gate->last) we = hipStreamWaitEvent(s, turn->gate->last, 0);
            if (*turn->abort_flag || we != hipSuccess) {
                if (turn->passed) *turn->passed = true;
                turn->gate->next = turn->seq + 1;
                turn->gate->cv.notify_all();
                return we != hipSuccess ? fail(e, BSW_E_HIP, "hipStreamWaitEvent: %s", hipGetErrorString(we)) : fail(e, BSW_E_HIP, "aborted: another chunk failed");
            }
        }
        hipError_t ce = hipSuccess;
        if (ci.packed) {               /* the words are the device layout already: they land in `seq`, nothing is packed */
            if (ci.words) ce = hipMemcpyAsync(st.d_seq.p, ci.direct ? (const void *)ci.lo : (const void *)st.h_raw.p, ci.words * 8, hipMemcpyHostToDevice, s);
        } else if (rawb) ce = hipMemcpyAsync(st.d_raw.p + RAW_FRONT, ci.direct ? ci.lo : st.h_raw.p, rawb, hipMemcpyHostToDevice, s);
        if (ce == hipSuccess) ce = hipMemcpyAsync(st.d_tasks.p, st.h_tasks.p, n * sizeof(bsw_dtask), hipMemcpyHostToDevice, s);
        if (ce == hipSuccess && (!ci.packed || ci.rb.use_ro)) ce = hipMemcpyAsync(st.d_roff.p, st.h_roff.p, n * sizeof(bsw_rawoff), hipMemcpyHostToDevice, s);
        if (ce == hipSuccess && n_desc) ce = hipMemcpyAsync(st.d_desc.p, st.h_desc.p, n_desc * sizeof(bsw_refx), hipMemcpyHostToDevice, s);
        if (turn) {
            if (ce == hipSuccess) ce = hipEventRecord(turn->ev, s);
            if (ce == hipSuccess) turn->gate->last = turn->ev;
            if (turn->passed) *turn->passed = true;
            turn->gate->next = turn->seq + 1;          /* pass the turn on even on failure: nobody may wait forever */
            turn->gate->cv.notify_all();
        }
        if (ce != hipSuccess) return fail(e, BSW_E_HIP, "input DMA: %s", hipGetErrorString(ce));
    }
    (void)dev_targets;
    if (ci.rb_on) HIPCHK(e, bsw::launch_rebase(st.d_tasks.p, st.d_roff.p, (uint32_t)n, ci.rb, s));
    if (!ci.packed)
        HIPCHK(e, bsw::launch_pack(st.d_raw.p + RAW_FRONT, st.d_tasks.p, st.d_roff.p, ci.raw_bias, (uint32_t)n, ci.rev_left ? 1 : 0,
                                   ref ? ref->d_pac[dev_index] : nullptr, ref ? ref->l_pac : 0, ref ? st.d_desc.p : nullptr, st.d_seq.p, st.d_nflag.p, s));
    HIPCHK(e, bsw::launch_bin(ci.bp, st.d_seq.p, ci.packed ? nullptr : st.d_nflag.p, st.d_tasks.p, (uint32_t)n, st.d_bins.p, st.d_keys.p, st.d_order.p, s));
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
         * (class, query with / without an N, query length descending, left sides: h0 bucket ascending); the order inside one
         * bin is task order here, arbitrary there */
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
            if (bp.nsplit && bits == 8 && (has_n(tasks[i].lquery, T.lqlen) | has_n(tasks[i].rquery, T.rqlen))) { k0 = BSW_BIN_NLIST; return; }
            const bool fz = bp.fused && bits == 8;
            if (fz) k1 = BSW_BIN_L(0, has_n(tasks[i].lquery, T.lqlen) | (bp.fused == 1 ? has_n(tasks[i].rquery, T.rqlen) : 0), bsw_h0_bucket(&bp, T.h0), T.lqlen);
            else if (T.lqlen) k1 = BSW_BIN_L(bits == 16, has_n(tasks[i].lquery, T.lqlen), bsw_h0_bucket(&bp, T.h0), T.lqlen);
            if (T.rqlen && !fz) k2 = BSW_BIN_R(bits == 16, has_n(tasks[i].rquery, T.rqlen), T.rqlen);
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
                const int bits = bp.lane_bits[c], b16 = bits == 16;
                for (int hn = 1; hn >= 0; --hn)
                    for (int q = BSW_LANE_QBINS - 1; q >= 0; --q) {
                        if (bsw_side_lane_class(&bp, bits, q) != c) continue;
                        for (int hb = 0; hb < (side ? 1 : BSW_H0_BUCKETS); ++hb) {
                            const size_t idx = (size_t)(side ? BSW_BIN_R(b16, hn, q) : BSW_BIN_L(b16, hn, hb, q));
                            cur[idx] = run;
                            run += hist[idx];
                        }
                    }
            }
        for (int c = 0; c < bp.n_wave; ++c) cur[(size_t)(BSW_BIN_WAVE0 + c)] = bp.wave_start[c];
        cur[BSW_BIN_LANEALL] = bp.lane_all_off;
        cur[BSW_BIN_NLIST] = bp.nlist_off;
        if (bp.nsplit) for (uint32_t q = 0; q < bp.fill_len; ++q) order[bp.fill_off + q] = BSW_ORDER_NONE;
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
    delete b->ci;
    delete b;
}

static void fill_refx(const bsw_ref_task *rt, size_t n, bsw_refx *x);

static int upload_common(bsw_ctx *ctx, const bsw_params *p, const bsw_task *tasks, size_t n, bsw_dev_batch **out,
                         const bsw_ref *ref /* NULL: targets come from the host */, const bsw_ref_task *rtasks, bool packed = false,
                         bool keep_raw = false)
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
        if (packed) gather_packed(tasks, st.h_tasks.p, n, (uint64_t *)st.h_raw.p, ci.rb);
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
    if (keep_raw) {                     /* bsw_upload_raw: raw bytes, offsets, bin scratch (and the target coordinates) stay */
        b->staged = true;
        b->ci = new chunk_info(ci);
        b->ref = ref;
        st.d_woff.release(); st.d_blob.release(); st.d_wout.release();
    } else
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

/* bsw_upload that KEEPS what crossed PCIe in HBM (byte-per-base sequences, their offsets) next to the packed form, so that
 * bsw_run_staged can re-run the device side of the batch manager — pack + bin — in front of the DP kernels: the whole hot
 * path from "the caller's bytes are in HBM" on, as one timed step. */
extern "C" int bsw_upload_raw(bsw_ctx *ctx, const bsw_params *p, const bsw_task *tasks, size_t n, bsw_dev_batch **out)
{
    if (!ctx) return BSW_E_INVAL;
    if (!out || (!tasks && n)) return fail(ctx->err, BSW_E_INVAL, "bsw_upload_raw: NULL argument");
    int rc = busy_check(ctx, "bsw_upload_raw");
    if (rc) return rc;
    return upload_common(ctx, p, tasks, n, out, nullptr, nullptr, false, true);
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
BSW_LOCAL int ref_to_task(errs &e, const bsw_params *p, int64_t l_pac, const bsw_ref_task &r, size_t i, bool rev_left,
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

static int run_common(bsw_ctx *ctx, bsw_dev_batch *b, bool staged, const char *what)
{
    if (!ctx) return BSW_E_INVAL;
    errs &e = ctx->err;
    if (!b) return fail(e, BSW_E_INVAL, "%s: NULL argument", what);
    if (staged && (!b->staged || !b->ci)) return fail(e, BSW_E_INVAL, "%s: the batch was not uploaded with bsw_upload_raw", what);
    int rc = busy_check(ctx, what);
    if (rc) return rc;
    HIPCHK(e, hipSetDevice(ctx->device0()));
    hipStream_t s = ctx->stream0();
    hipEvent_t e0 = ctx->ev_start, e1 = ctx->ev_stop, mid = nullptr;
    if (ctx->hist_used < 4096) {
        if (ctx->hist_used == ctx->hist.size()) {
            bsw_ctx::run_ev r;
            HIPCHK(e, hipEventCreate(&r.e0));
            HIPCHK(e, hipEventCreate(&r.mid));
            HIPCHK(e, hipEventCreate(&r.e1));
            ctx->hist.push_back(r);
        }
        bsw_ctx::run_ev &r = ctx->hist[ctx->hist_used];
        e0 = r.e0; e1 = r.e1; mid = r.mid;
        r.staged = staged;
        ++ctx->hist_used;
    }
    HIPCHK(e, hipEventRecord(e0, s));
    b->launches = 0;
    if (staged && b->n) {
        const chunk_info &ci = *b->ci;
        stage_t &st = b->st;
        if (!ci.packed) {
            HIPCHK(e, bsw::launch_pack(st.d_raw.p + RAW_FRONT, st.d_tasks.p, st.d_roff.p, ci.raw_bias, (uint32_t)b->n, ci.rev_left ? 1 : 0,
                                       nullptr, 0, nullptr, st.d_seq.p, st.d_nflag.p, s));
            ++b->launches;
        }
        HIPCHK(e, bsw::launch_bin(ci.bp, st.d_seq.p, ci.packed ? nullptr : st.d_nflag.p, st.d_tasks.p, (uint32_t)b->n, st.d_bins.p, st.d_keys.p, st.d_order.p, s));
        b->launches += 3;
    }
    if (mid) HIPCHK(e, hipEventRecord(mid, s));
    rc = enqueue_batch(e, b->P, b->variant, b->st.d_seq.p, b->st.d_tasks.p, b->st.d_order.p, b->plan, b->st.d_out.p, s, &b->launches, fork_for(ctx, s));
    if (rc) return rc;
    HIPCHK(e, hipEventRecord(e1, s));
    ctx->ev_last0 = e0; ctx->ev_last1 = e1;
    ctx->timed = true;
    return BSW_OK;
}

extern "C" int bsw_run(bsw_ctx *ctx, bsw_dev_batch *b) { return run_common(ctx, b, false, "bsw_run"); }

/* the device side of the batch manager (pack: byte per base -> 16 bases per uint64; bin: the counting sort into launch
 * lists) AND the DP kernels of a batch uploaded with bsw_upload_raw, on the library's stream: what a chunk of bsw_submit
 * runs on the GPU once its bytes have landed (tbb.v:110-123 -> sw_pe_array_task_parse.v:1600-1648 -> the PEs) */
extern "C" int bsw_run_staged(bsw_ctx *ctx, bsw_dev_batch *b) { return run_common(ctx, b, true, "bsw_run_staged"); }

extern "C" int bsw_sync(bsw_ctx *ctx)
{
    if (!ctx) return BSW_E_INVAL;
    errs &e = ctx->err;
    if (pipeline_busy(ctx)) return fail(e, BSW_E_BUSY, "bsw_sync: a bsw_submit is in flight (call bsw_wait)");
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

/* total_ms[k] = the k-th run since the last call (first event to last); staging_ms[k] (may be NULL) = its pack + bin part,
 * 0 for a bsw_run */
extern "C" int bsw_run_history2(bsw_ctx *ctx, float *total_ms, float *staging_ms, int cap)
{
    if (!ctx || (!total_ms && cap > 0)) return BSW_E_INVAL;
    errs &e = ctx->err;
    int rc = bsw_sync(ctx);
    if (rc) return rc;
    int n = 0;
    for (size_t i = 0; i < ctx->hist_used && n < cap; ++i, ++n) {
        const bsw_ctx::run_ev &r = ctx->hist[i];
        HIPCHK(e, hipEventElapsedTime(&total_ms[n], r.e0, r.e1));
        if (staging_ms) {
            staging_ms[n] = 0.f;
            if (r.staged) HIPCHK(e, hipEventElapsedTime(&staging_ms[n], r.e0, r.mid));
        }
    }
    ctx->hist_used = 0;
    return n;
}

extern "C" int bsw_run_history(bsw_ctx *ctx, float *ms, int cap) { return bsw_run_history2(ctx, ms, nullptr, cap); }

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

extern "C" int bsw_chain_timeouts(bsw_ctx *ctx, uint64_t *n)
{
    if (!ctx || !n) return BSW_E_INVAL;
    errs &e = ctx->err;
    int rc = bsw_sync(ctx);
    if (rc) return rc;
    uint64_t total = 0;
    for (dev_state &d : ctx->devs)
        for (fork_t &f : d.forks)
            if (f.ok && f.mode == 2 && f.flag_mem) {
                uint32_t v = 0;
                HIPCHK(e, hipSetDevice(d.device));
                HIPCHK(e, hipMemcpy(&v, f.expired(), sizeof(v), hipMemcpyDeviceToHost));
                total += v;
            }
    HIPCHK(e, hipSetDevice(ctx->device0()));
    *n = total;
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
/* an enqueue failed half-way: wait for what IS in flight (DMAs out of the pinned staging, kernels storing into the caller's
 * buffers) before the caller may reuse or free anything; a wait that fails too marks the context dead (sync_stream does) */
static void drain_after_error(bsw_ctx *ctx, errs &e, hipStream_t s, hipEvent_t ev)
{
    errs keep = e;
    if (sync_stream(ctx, e, s, ev) != BSW_OK) ctx->dead = true;      /* nothing can be promised about the buffers any more */
    else e = keep;                                                   /* report the launch failure, not the drain */
}
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
    HIPCHK(e, hipMemcpyAsync(st.d_blob.p, hb, total * sizeof(uint64_t), hipMemcpyHostToDevice, s));      /* (the first thing on the stream: nothing to drain if it fails) */
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
        if (!cnt) continue;
        if (hipError_t le = bsw::launch_wave(c, p.variant, dp, d_seq, d_tasks, d_order + pl.wave_start[c], cnt, nullptr, ctr + 1 + c, res, s)) {
            /* kernels already launched store into pinned h_out / the caller's registered `out`, and the H2D copy of h_blob may
             * still be reading it: nothing of this call may be in flight when the caller gets its buffers back */
            drain_after_error(ctx, e, s, ev);
            return fail(e, BSW_E_HIP, "launch: %s", hipGetErrorString(le));
        }
    }
    rc = sync_stream(ctx, e, s, ev);
    if (rc) return rc;
    if (!out_direct) memcpy(out, st.h_out.p, n * sizeof(bsw_result));
    return BSW_OK;
}

/* ---- one synchronous chunk through a staging slot (small batches; the streaming workers use the same steps) ---- */
BSW_LOCAL int run_chunk(bsw_ctx *ctx, errs &e, stage_t &st, hipStream_t s, hipEvent_t ev, const bsw_params &p, const bsw_dparams &dp,
                     const bsw_task *tasks, size_t n, bsw_result *out, int gather_threads, const gate_turn *turn, bool packed)
{
    if (n == 0) return BSW_OK;
    {
        static const bool nosmall = getenv("BSW_NO_SMALL") != nullptr;     /* (measurements) */
        if (!nosmall && n <= SMALL_BATCH && !packed && !turn && ctx->cfg.kernel != BSW_KERNEL_LANE)
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
        if (packed) gather_packed(tasks, st.h_tasks.p, n, (uint64_t *)st.h_raw.p, ci.rb);
        else gather_raw(tasks, st.h_roff.p, n, false, st.h_raw.p, gather_threads);
    }
    const double t_c = dbg ? tnow() : 0;
    rc = stage_device(e, st, s, ci, n, false, nullptr, nullptr, turn);
    if (rc) return rc;
    /* `out` is ALWAYS bsw_result[n] here, whatever the context's result_format: the only caller is bsw_extend_batch
     * (ext_group reads res[k].right); BSW_RESULT_PAIR applies to the bsw_submit* calls, whose slot workers have their own path */
    const size_t rec = sizeof(bsw_result);
    rc = enqueue_batch(e, dp, p.variant, st.d_seq.p, st.d_tasks.p, st.d_order.p, ci.plan, st.d_out.p, s, nullptr, fork_for(ctx, s), nullptr);
    if (rc) { drain_after_error(ctx, e, s, ev); return rc; }
    const bool out_direct = is_registered(out, n * rec);
    if (!out_direct && (he = st.h_out.reserve(n)) != hipSuccess) return fail(e, BSW_E_NOMEM, "pinned staging: %s", hipGetErrorString(he));
    if (hipError_t ce = hipMemcpyAsync(out_direct ? (void *)out : (void *)st.h_out.p, (const void *)st.d_out.p, n * rec, hipMemcpyDeviceToHost, s)) {
        drain_after_error(ctx, e, s, ev);
        return fail(e, BSW_E_HIP, "result DMA: %s", hipGetErrorString(ce));
    }
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


/* ---- the pipeline behind bsw_submit / bsw_submit_packed / bsw_submit_ref -------------------------------------------------
 * The reference keeps FOUR task batches in flight under one manager (batch_manager.v:343-348 request bits, :434-435 busy
 * bitmap) and its host polls a status word for completion (DSM + 0x40 busy nibble, :844-854).  Here: up to BSW_MAX_INFLIGHT
 * submits per context, each a TICKET; the chunks of all of them go through ONE queue per device in submit order, popped by
 * the device's slot threads — persistent, started by the first submit, pinned next to the card — so that the last chunks of
 * submit k and the first ones of submit k+1 overlap exactly as two neighbouring chunks of one submit do.
 * One slot = one host thread + one stream + one set of staging buffers.  Per chunk: host pass (validate, lay out, count) ->
 * wait for the slot's previous chunk -> input DMAs in the device's chunk order -> pack, bin, DP kernels -> result DMA.  The
 * host pass of the slot's next chunk runs while its previous one is still on the GPU: it only needs the pinned host staging,
 * which is free again as soon as that chunk's input DMAs are done. */
struct ticket_t {
    uint64_t id = 0;
    bsw_params p{};
    bsw_dparams dp{};
    const bsw_task *tasks = nullptr;
    const bsw_ref *ref = nullptr;
    const bsw_ref_task *rtasks = nullptr;      /* non-NULL: seeds against the device-resident reference */
    bsw_result *out = nullptr;
    bool packed = false;
    size_t n = 0;
    std::atomic<size_t> remaining{0};           /* chunks whose results have not been handed over yet */
    size_t nchunks = 0;                         /* chunks the submit was cut into */
    std::atomic<int> abort{0};                  /* a chunk failed: the others do nothing any more */
    std::mutex emu;
    int rc = 0;                                 /* the failure itself, not the chunks it made give up */
    bool real = false;
    errs err;
    bool done = false;                          /* (pipeline::mu) */
};

struct chunk_job {
    ticket_t *t = nullptr;
    chunk_span span{0, 0};
    size_t seq = 0;                             /* the chunk's place in its device's input-DMA order */
};

struct dev_pipe {
    std::deque<chunk_job> q;
    h2d_gate gate;
    size_t next_seq = 0;
};

struct pipeline {
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::vector<std::unique_ptr<dev_pipe>> devs;
    std::vector<std::thread> threads;
    std::deque<std::unique_ptr<ticket_t>> live; /* submits not collected by a wait yet, oldest first */
    uint64_t next_id = 1;
    bool stop = false;
    /* what the host side costs (bsw_host_stats): CPU time of the slot threads and of the gather helpers they start, volume */
    std::atomic<uint64_t> slot_cpu_ns{0}, helper_cpu_ns{0}, seeds{0}, chunks{0}, h2d_bytes{0}, d2h_bytes{0}, submits{0};
};

static void slot_main(bsw_ctx *ctx, size_t d, size_t s)
{
    pipeline &pp = *ctx->pipe;
    dev_pipe &dq = *pp.devs[d];
    dev_state &dev = ctx->devs[d];
    stage_t &st = dev.slots[s];
    hipStream_t stream = dev.streams[s];
    static const bool dbg = getenv("BSW_DEBUG_TIMING") != nullptr;
    auto tnow = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    /* the slot's thread (and the gather threads it starts, which inherit the mask) next to its GPU: the host pass streams the
     * caller's task array into write-combined pinned staging, and that staging is allocated — first touched — from here */
    pin_this_thread(dev);
    const hipError_t dev_err = hipSetDevice(dev.device);
    /* BSW_RESULT_PAIR: `out` addresses bsw_pair[n]; the dense 32-byte records come from their own device array */
    const bool pairs = ctx->cfg.result_format == BSW_RESULT_PAIR;
    const size_t rec = pairs ? sizeof(bsw_pair) : sizeof(bsw_result);
    auto d_res = [&]() -> const void * { return pairs ? (const void *)st.d_pair.p : (const void *)st.d_out.p; };
    struct { bool active = false; ticket_t *t = nullptr; size_t n = 0; char *out = nullptr; bool direct = false, copied = false; } pend;
    std::vector<bsw_task> rt_tasks;                 /* ref mode: this chunk's seeds as tasks (left queries by reference) */
    uint64_t cpu_last = thread_cpu_ns();
    auto account = [&]() { const uint64_t c = thread_cpu_ns(); pp.slot_cpu_ns += c - cpu_last; cpu_last = c; };

    auto ticket_fail = [&](ticket_t *t, int rc, const errs &er) {
        std::lock_guard<std::mutex> lk(t->emu);
        const bool real = er.msg.compare(0, 7, "aborted") != 0;
        if (!t->rc || (real && !t->real)) { t->rc = rc; t->err = er; t->real = real; }
        t->abort = 1;
    };
    auto chunk_done = [&](ticket_t *t) {
        if (t->remaining.fetch_sub(1) == 1) {
            std::lock_guard<std::mutex> lk(pp.mu);
            t->done = true;
            pp.cv_done.notify_all();
        }
    };
    auto finish = [&]() {                           /* the slot's chunk in flight: wait (watchdog), hand the results over */
        if (!pend.active) return;
        pend.active = false;
        errs fe;
        int rc = BSW_OK;
        if (!pend.copied) {                         /* kernels done -> result DMA -> done */
            rc = wait_event(ctx, fe, dev.events[s]);
            if (!rc) {
                const hipError_t ce = hipMemcpyAsync(pend.direct ? (void *)pend.out : (void *)st.h_out.p, d_res(), pend.n * rec, hipMemcpyDeviceToHost, stream);
                if (ce != hipSuccess) rc = fail(fe, BSW_E_HIP, "result DMA: %s", hipGetErrorString(ce));
            }
        }
        if (!rc) rc = sync_stream(ctx, fe, stream, dev.events[s]);
        else { errs quiet; (void)sync_stream(ctx, quiet, stream, dev.events[s]); }     /* leave nothing of the chunk in flight */
        if (!rc && !pend.direct) memcpy(pend.out, st.h_out.p, pend.n * rec);
        if (rc) ticket_fail(pend.t, rc, fe);
        else { pp.d2h_bytes += pend.n * rec; pp.seeds += pend.n; }
        chunk_done(pend.t);
    };
    auto process = [&](const chunk_job &job) {
        ticket_t *t = job.t;
        const size_t n = job.span.cnt, base = job.span.base;
        const bsw_params &p = t->p;
        const bsw_task *ct = t->tasks ? t->tasks + base : nullptr;
        errs e;
        int rc = BSW_OK;
        hipError_t he;
        chunk_info ci;
        ci.streaming = t->nchunks >= 3;
        bool passed = false, queued = false;         /* the DMA turn has been passed on; some async op of this chunk may be on the stream */
        const double t0 = dbg ? tnow() : 0;
        double t1 = t0, t2 = t0;
        /* the host pass of a chunk on pass_threads threads side by side (this one + helpers started for the pass): the context's
         * pack_threads spread over its slots, at least two */
        const int pass_threads = std::max(2, ctx->cfg.pack_threads / (int)dev.slots.size() + 1);
        do {
            if (dev_err != hipSuccess) { rc = fail(e, BSW_E_HIP, "hipSetDevice: %s", hipGetErrorString(dev_err)); break; }
            if (t->abort) { rc = fail(e, BSW_E_HIP, "aborted: another chunk failed"); break; }
            if (pend.active && (rc = wait_event(ctx, e, dev.h2d_done[s]))) break;           /* pinned host staging is free again */
            t1 = dbg ? tnow() : 0;
            if ((he = st.h_tasks.reserve(n + 1)) != hipSuccess || (he = st.h_roff.reserve(n + 1)) != hipSuccess ||
                (t->rtasks && (he = st.h_desc.reserve(n + 1)) != hipSuccess)) { rc = fail(e, BSW_E_NOMEM, "pinned staging: %s", hipGetErrorString(he)); break; }
            if (t->rtasks) {                        /* mem_chain2aln's task extraction fused into the pass; the targets stay on the device */
                const bsw_ref_task *crt = t->rtasks + base;
                bsw_refx *rx = st.h_desc.p;
                size_t so = 0;
                std::mutex emu;                      /* (the pass runs on several threads: a failing seed's text is written under it) */
                rc = prepare_chunk_t(e, &p, ctx->cfg.kernel, [&](size_t i, bsw_task &tmp, int &erc) -> const bsw_task * {
                    errs le;
                    size_t so_l = 0;                 /* (rev_left: nothing is written to scratch) */
                    erc = ref_to_task(le, &p, t->ref->l_pac, crt[i], base + i, true, nullptr, so_l, tmp);
                    if (erc) { std::lock_guard<std::mutex> lk(emu); e = le; }
                    rx[i] = bsw_refx{crt[i].seed.rbeg - 1, crt[i].seed.rbeg + crt[i].seed.len};
                    return erc ? nullptr : &tmp;
                }, n, true, st.h_tasks.p, st.h_roff.p, ci, true, false, base, pass_threads, &pp.helper_cpu_ns);
                if (rc) break;
                if (!ci.direct) {                   /* reads in pageable memory: materialise the tasks for the gather */
                    rt_tasks.resize(n);
                    for (size_t i = 0; i < n && !rc; ++i) rc = ref_to_task(e, &p, t->ref->l_pac, crt[i], base + i, true, nullptr, so, rt_tasks[i]);
                    if (rc) break;
                    ct = rt_tasks.data();
                    gather_offsets(ct, n, true, st.h_roff.p);
                }
            } else if ((rc = prepare_chunk(e, &p, ctx->cfg.kernel, ct, n, false, st.h_tasks.p, st.h_roff.p, ci, false, t->packed, base, pass_threads, &pp.helper_cpu_ns))) break;
            if (!ci.direct) {
                if ((he = st.h_raw.reserve(ci.sum_len + RAW_SLACK)) != hipSuccess) { rc = fail(e, BSW_E_NOMEM, "pinned staging: %s", hipGetErrorString(he)); break; }
                if (t->packed) gather_packed(ct, st.h_tasks.p, n, (uint64_t *)st.h_raw.p, ci.rb);
                else gather_raw(ct, st.h_roff.p, n, t->rtasks != nullptr, st.h_raw.p, std::max(1, ctx->cfg.pack_threads / (int)dev.slots.size()), t->rtasks != nullptr, &pp.helper_cpu_ns);
            }
        } while (0);
        t2 = dbg ? tnow() : 0;
        if (rc) ticket_fail(t, rc, e);              /* (before the previous chunk reports what this failure did to the context) */
        finish();                                   /* the slot's previous chunk, whatever became of this one */
        const double t3 = dbg ? tnow() : 0;
        char *co = (char *)t->out + base * rec;
        uint64_t h2d = 0;
        if (!rc) {
            gate_turn turn;
            turn.gate = &dq.gate; turn.seq = job.seq; turn.ev = dev.h2d_done[s]; turn.abort_flag = &t->abort; turn.passed = &passed;
            queued = true;
            rc = stage_device(e, st, stream, ci, n, t->rtasks != nullptr, t->ref, &h2d, &turn, d);
            if (!rc && pairs && (he = st.d_pair.reserve(n + 1)) != hipSuccess) rc = fail(e, BSW_E_NOMEM, "device staging: %s", hipGetErrorString(he));
            if (!rc) rc = enqueue_batch(e, t->dp, p.variant, st.d_seq.p, st.d_tasks.p, st.d_order.p, ci.plan, st.d_out.p, stream, nullptr, fork_for(ctx, stream, true), pairs ? st.d_pair.p : nullptr);
        }
        if (!rc) {
            pend.direct = is_registered(co, n * rec);
            if (!pend.direct && (he = st.h_out.reserve(n)) != hipSuccess) rc = fail(e, BSW_E_NOMEM, "pinned staging: %s", hipGetErrorString(he));
        }
        if (!rc) {
            /* A result copy queued behind its kernels sits at the head of its DMA engine's ring until they finish and holds up
             * the copies queued to that engine after it (profiles/r2/wire_submit_timeline.txt): the slot thread issues the result
             * DMA itself once the chunk's kernels are done (it waits for them anyway before it reuses the staging).
             * profiles/r5/e2e_late_result_dma.txt: the byte path loses its outliers (max / min 1.05 - 1.07 instead of 1.2 - 1.8), a
             * stream of packed submits goes from 110 - 113 to 128 - 134 M seeds/s.  BSW_LATE_RESULT=0: the copy is queued right away. */
            static const int late_env = getenv("BSW_LATE_RESULT") ? atoi(getenv("BSW_LATE_RESULT")) : -1;     /* (measurements) */
            const bool late = late_env != 0;
            if (!late) he = hipMemcpyAsync(pend.direct ? (void *)co : (void *)st.h_out.p, d_res(), n * rec, hipMemcpyDeviceToHost, stream);
            else he = hipEventRecord(dev.events[s], stream);
            if (he != hipSuccess) rc = fail(e, BSW_E_HIP, "result DMA: %s", hipGetErrorString(he));
            else {
                pend.copied = !late;
                pend.active = true; pend.t = t; pend.n = n; pend.out = co;
                pp.h2d_bytes += h2d;
                pp.chunks += 1;
            }
        }
        if (rc) {
            /* the turn is passed on whatever happened (chunks of other submits queue behind this one); a failure after the
             * first input DMA was queued leaves copies out of the caller's registered arena and kernels in flight: drain the
             * stream (with the watchdog) before the failure is reported, so the caller may free or reuse that memory as soon
             * as the wait returns */
            if (!passed) {
                std::unique_lock<std::mutex> lk(dq.gate.mu);
                dq.gate.cv.wait(lk, [&]() { return dq.gate.next == job.seq; });
                dq.gate.next = job.seq + 1;
                dq.gate.cv.notify_all();
            }
            if (queued) { errs quiet; (void)sync_stream(ctx, quiet, stream, dev.events[s]); }
            ticket_fail(t, rc, e);
            chunk_done(t);
        }
        if (dbg) fprintf(stderr, "[bsw] slot %zu.%zu ticket %llu chunk @%zu n=%zu: wait staging %.3f ms, host pass %.3f, wait prev results %.3f, DMA turn + enqueue %.3f (t0=%.3f)%s\n",
                         d, s, (unsigned long long)t->id, base, n, t1 - t0, t2 - t1, t3 - t2, tnow() - t3, t0, rc ? " FAILED" : "");
    };

    std::unique_lock<std::mutex> lk(pp.mu);
    for (;;) {
        if (!dq.q.empty()) {
            const chunk_job job = dq.q.front();
            dq.q.pop_front();
            lk.unlock();
            process(job);
            account();
            lk.lock();
            continue;
        }
        if (pend.active) {                          /* nothing queued behind it: hand the chunk in flight over now */
            lk.unlock();
            finish();
            account();
            lk.lock();
            continue;
        }
        if (pp.stop) break;
        pp.cv_work.wait(lk);
        cpu_last = thread_cpu_ns();
    }
}

static int pipeline_start(bsw_ctx *ctx)
{
    if (ctx->pipe) return BSW_OK;
    pipeline *pp = new pipeline();
    const size_t G = ctx->devs.size(), S = (size_t)ctx->cfg.streams;
    for (size_t d = 0; d < G; ++d) pp->devs.emplace_back(new dev_pipe());
    ctx->pipe = pp;
    for (size_t s = 0; s < S; ++s)
        for (size_t d = 0; d < G; ++d) pp->threads.emplace_back(slot_main, ctx, d, s);
    return BSW_OK;
}

BSW_LOCAL bool pipeline_busy(bsw_ctx *ctx)
{
    if (!ctx->pipe) return false;
    std::lock_guard<std::mutex> lk(ctx->pipe->mu);
    return !ctx->pipe->live.empty();
}

BSW_LOCAL void pipeline_shutdown(bsw_ctx *ctx)
{
    pipeline *pp = ctx->pipe;
    if (!pp) return;
    {
        std::unique_lock<std::mutex> lk(pp->mu);
        pp->cv_done.wait(lk, [&]() { for (auto &t : pp->live) if (!t->done) return false; return true; });
        pp->live.clear();
        pp->stop = true;
    }
    pp->cv_work.notify_all();
    for (auto &t : pp->threads) t.join();
    ctx->pipe = nullptr;
    delete pp;
}

static int submit_common(bsw_ctx *ctx, const bsw_params *p, const bsw_task *tasks, const bsw_ref *ref, const bsw_ref_task *rtasks,
                         size_t n, bsw_result *out, bool packed, bsw_ticket *ticket, const char *what)
{
    if (ticket) *ticket = 0;
    if (ctx->dead) return fail(ctx->err, BSW_E_HIP, "%s: context is dead (an earlier wait for the GPU timed out)", what);
    bsw_dparams dp;
    int rc = check_params(ctx->err, p, &dp);
    if (rc) return rc;
    if ((rc = pipeline_start(ctx))) return rc;
    pipeline &pp = *ctx->pipe;
    const size_t G = ctx->devs.size();
    std::unique_ptr<ticket_t> t(new ticket_t());
    t->p = *p; t->dp = dp; t->tasks = tasks; t->ref = ref; t->rtasks = rtasks; t->out = out; t->packed = packed; t->n = n;
    /* chunk_tasks = 0 (the default): sized by WORK, not by seeds.  A chunk's launches must fill the machine — 128 Ki seeds of
     * the 150 bp single bin (131-base sides) do; PE seeds with their two shorter sides hold half the cells per seed and want
     * twice the seeds (sweep on 4 M PE seeds, profiles/r6/e2e_pe_chunk_sweep.txt: packed input 67 / 91 / 104 / 86 M seeds/s
     * for 128 / 192 / 256 / 384 Ki).  Work of a seed ~ lqlen^2 + rqlen^2 (rows x live columns both grow with the side), taken
     * over a strided sample of the submit. */
    size_t chunk = ctx->cfg.chunk_tasks;
    if (chunk == 0) {
        chunk = 131072;
        if (n > 0) {
            const size_t step = std::max<size_t>(1, n / 2048);
            double acc = 0;
            size_t cnt = 0;
            for (size_t i = 0; i < n; i += step, ++cnt) {
                double l, r;
                if (rtasks) { l = rtasks[i].seed.qbeg; r = rtasks[i].l_query - rtasks[i].seed.qbeg - rtasks[i].seed.len; }
                else { l = tasks[i].lqlen; r = tasks[i].rqlen; }
                acc += l * l + r * r;
            }
            const double mean = cnt ? acc / (double)cnt : 0.0;
            /* ... and a big submit takes chunks of twice that: a launch that fills the machine twice over loses less to its
             * longest waves (2 M resident PE seeds as chunks on two streams, no PCIe: 128 / 256 / 512 Ki -> 85 / 160 / 219 M
             * seeds/s against 240 as one batch, profiles/r6/resident_chunks_pe.txt), as long as ~16 chunks are left to pipeline */
            double base = 131072.0;
            if (mean > 1.0) base = std::min(524288.0, std::max(65536.0, 131072.0 * (131.0 * 131.0) / mean));
            chunk = (size_t)std::min(2.0 * base, std::max(base, (double)n / 16.0));
            chunk = (chunk + 8191) & ~(size_t)8191;
        }
    }
    const std::vector<std::vector<chunk_span>> chunks = plan_chunks(n, chunk, G);
    size_t total = 0;
    for (const auto &v : chunks) total += v.size();
    t->remaining = total;
    t->nchunks = total;
    t->done = total == 0;
    {
        std::lock_guard<std::mutex> lk(pp.mu);
        if (pp.live.size() >= BSW_MAX_INFLIGHT)
            return fail(ctx->err, BSW_E_BUSY, "%s: %d submits in flight already (BSW_MAX_INFLIGHT); wait for one first", what, BSW_MAX_INFLIGHT);
        t->id = pp.next_id++;
        if (ticket) *ticket = t->id;
        /* chunk c of the submit -> device c mod G; the devices' queues are filled in the submit's own chunk order */
        size_t left = total;
        for (size_t k = 0; left; ++k)
            for (size_t d = 0; d < G; ++d)
                if (k < chunks[d].size()) { pp.devs[d]->q.push_back(chunk_job{t.get(), chunks[d][k], pp.devs[d]->next_seq++}); --left; }
        pp.live.push_back(std::move(t));
        pp.submits += 1;
    }
    pp.cv_work.notify_all();
    return BSW_OK;
}

extern "C" int bsw_submit_t(bsw_ctx *ctx, const bsw_params *p, const bsw_task *tasks, size_t n, bsw_result *out, bsw_ticket *ticket)
{
    if (!ctx) return BSW_E_INVAL;
    if (!p || (!tasks && n) || (!out && n)) return fail(ctx->err, BSW_E_INVAL, "bsw_submit: NULL argument");
    return submit_common(ctx, p, tasks, nullptr, nullptr, n, out, false, ticket, "bsw_submit");
}
extern "C" int bsw_submit(bsw_ctx *ctx, const bsw_params *p, const bsw_task *tasks, size_t n, bsw_result *out) { return bsw_submit_t(ctx, p, tasks, n, out, nullptr); }

/* bsw_submit for callers that keep their sequences 4-bit packed — 16 bases per uint64, base k in bits [4k, 4k+3], codes
 * 0-3 = ACGT, 4-7 = N, every sequence on an 8-byte boundary, lengths still in bases: the device's own layout and the
 * encoding the reference ships over its link (8 bases per 32-bit word, sw_pe_array_proc_element.v:1638,1677-1683).  The
 * words of a registered arena are DMA'd straight into the sequence buffer: no pack kernel, less than half the PCIe bytes
 * of byte-per-base input.  bsw_pack_bases() converts one sequence. */
extern "C" int bsw_submit_packed_t(bsw_ctx *ctx, const bsw_params *p, const bsw_task *tasks, size_t n, bsw_result *out, bsw_ticket *ticket)
{
    if (!ctx) return BSW_E_INVAL;
    if (!p || (!tasks && n) || (!out && n)) return fail(ctx->err, BSW_E_INVAL, "bsw_submit_packed: NULL argument");
    return submit_common(ctx, p, tasks, nullptr, nullptr, n, out, true, ticket, "bsw_submit_packed");
}
extern "C" int bsw_submit_packed(bsw_ctx *ctx, const bsw_params *p, const bsw_task *tasks, size_t n, bsw_result *out) { return bsw_submit_packed_t(ctx, p, tasks, n, out, nullptr); }

/* bsw_submit for seeds against a DEVICE-RESIDENT reference (F3): only the reads cross PCIe; the targets are fetched
 * from the 2-bit pac on the GPU, the left flank of every read is mirrored by the pack kernel. */
extern "C" int bsw_submit_ref_t(bsw_ctx *ctx, const bsw_params *p, const bsw_ref *ref, const bsw_ref_task *rtasks, size_t n, bsw_result *out, bsw_ticket *ticket)
{
    if (!ctx) return BSW_E_INVAL;
    if (!p || !ref || (!rtasks && n) || (!out && n)) return fail(ctx->err, BSW_E_INVAL, "bsw_submit_ref: NULL argument");
    if (ref->d_pac.size() != ctx->devs.size()) return fail(ctx->err, BSW_E_INVAL, "bsw_submit_ref: the reference was uploaded through another context");
    return submit_common(ctx, p, nullptr, ref, rtasks, n, out, false, ticket, "bsw_submit_ref");
}
extern "C" int bsw_submit_ref(bsw_ctx *ctx, const bsw_params *p, const bsw_ref *ref, const bsw_ref_task *rtasks, size_t n, bsw_result *out) { return bsw_submit_ref_t(ctx, p, ref, rtasks, n, out, nullptr); }

/* every submit in flight; the first failure in submit order is returned (and is what bsw_last_error describes) */
extern "C" int bsw_wait(bsw_ctx *ctx)
{
    if (!ctx) return BSW_E_INVAL;
    pipeline *pp = ctx->pipe;
    if (!pp) return BSW_OK;
    std::unique_lock<std::mutex> lk(pp->mu);
    pp->cv_done.wait(lk, [&]() { for (auto &t : pp->live) if (!t->done) return false; return true; });
    int rc = BSW_OK;
    for (auto &t : pp->live)
        if (t->rc && !rc) { rc = t->rc; ctx->err = t->err; }
    pp->live.clear();
    return rc;
}

extern "C" int bsw_wait_ticket(bsw_ctx *ctx, bsw_ticket ticket)
{
    if (!ctx) return BSW_E_INVAL;
    pipeline *pp = ctx->pipe;
    if (!pp) return fail(ctx->err, BSW_E_INVAL, "bsw_wait_ticket: no such ticket");
    std::unique_lock<std::mutex> lk(pp->mu);
    auto find = [&]() { for (size_t i = 0; i < pp->live.size(); ++i) if (pp->live[i]->id == ticket) return (long)i; return -1L; };
    long i = find();
    if (i < 0) return fail(ctx->err, BSW_E_INVAL, "bsw_wait_ticket: no such ticket (%llu)", (unsigned long long)ticket);
    ticket_t *t = pp->live[(size_t)i].get();
    pp->cv_done.wait(lk, [&]() { return t->done; });
    const int rc = t->rc;
    if (rc) ctx->err = t->err;
    pp->live.erase(pp->live.begin() + find());
    return rc;
}

/* the host's status poll (batch_manager.v:844-854): 1 = the submit is complete (results are in out[]; collect it — and its
 * error code, if any — with bsw_wait_ticket / bsw_wait), 0 = still in flight.  Never blocks. */
extern "C" int bsw_test(bsw_ctx *ctx, bsw_ticket ticket)
{
    if (!ctx) return BSW_E_INVAL;
    pipeline *pp = ctx->pipe;
    if (pp) {
        std::lock_guard<std::mutex> lk(pp->mu);
        for (auto &t : pp->live)
            if (t->id == ticket) return t->done ? 1 : 0;
    }
    return fail(ctx->err, BSW_E_INVAL, "bsw_test: no such ticket (%llu)", (unsigned long long)ticket);
}

extern "C" int bsw_inflight(bsw_ctx *ctx)
{
    if (!ctx) return BSW_E_INVAL;
    if (!ctx->pipe) return 0;
    std::lock_guard<std::mutex> lk(ctx->pipe->mu);
    return (int)ctx->pipe->live.size();
}

extern "C" int bsw_host_stats(bsw_ctx *ctx, bsw_stats *out, size_t out_size)
{
    if (!ctx || !out || out_size < sizeof(uint64_t)) return BSW_E_INVAL;
    bsw_stats s;
    memset(&s, 0, sizeof(s));
    if (pipeline *pp = ctx->pipe) {
        s.slot_cpu_ns = pp->slot_cpu_ns; s.helper_cpu_ns = pp->helper_cpu_ns; s.seeds = pp->seeds; s.chunks = pp->chunks;
        s.h2d_bytes = pp->h2d_bytes; s.d2h_bytes = pp->d2h_bytes; s.submits = pp->submits; s.slot_threads = pp->threads.size();
    }
    memcpy(out, &s, std::min(out_size, sizeof(s)));
    return BSW_OK;
}
