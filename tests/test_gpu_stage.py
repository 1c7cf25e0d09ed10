"""GPU tests of the round-2 batch manager: device-side packing and binning, DMA straight out of registered
host memory, several devices behind one context, host-supplied band limits (H5/H6), queued wire-format
batches, the coalescing scalar ABI and the watchdog.  Parity is always against the CPU oracle."""
import ctypes as C
import threading

import numpy as np
import pytest

import _gen
from test_gpu_parity import assert_same, FIELDS

pytestmark = pytest.mark.gpu

MIXED = dict(seed_len_min=19, seed_len_max=60, seed_at_start=0, sub_rate=0.02, indel_rate=0.01, junk_frac=0.1, n_rate=0.001)


def rebase(tasks, arena, dst_u8, content=None):
    """Copy `content` (default: the arena itself) into dst_u8 and return tasks whose pointers follow."""
    src = arena if content is None else content
    dst_u8[:src.size] = src
    delta = dst_u8.ctypes.data - arena.ctypes.data
    t = tasks.copy()
    for f in ("lquery", "ltarget", "rquery", "rtarget"):
        nz = t[f] != 0
        t[f][nz] = (t[f][nz].astype(np.int64) + delta).astype(np.uint64)
    return t


@pytest.mark.parametrize("kernel", [0, 1, 2])
def test_device_binning_matches_the_plan(host, kernel):
    """bsw_bin_* on the GPU must produce the lists bsw_plan_batch promises: same segments, every list a
    permutation of the host replay's, lane sides sorted by (query holds an N, query length descending)."""
    n = 2 * host.LANE_AUTO_MIN + 7000           # both sides are launched: the threshold counts per side
    tasks, arena = host.synth_tasks(n, seed=5, read_len=250, seed_len_min=19, seed_len_max=120, seed_at_start=0,
                                    junk_frac=0.1, n_rate=0.001)
    tasks["h0"][::7] = 300                     # some seeds outside the 8-bit score range
    p = host.default_params()
    want_order, want_seg, _ = host.plan_batch(p, tasks, kernel=kernel)
    with host.BswContext(device=0, kernel=kernel) as c:
        b = c.upload(p, tasks)
        order, seg = c.batch_order(b)
        b.free()
    assert (seg == want_seg).all()
    for s in range(25):
        lo, hi = int(seg[s]), int(seg[s + 1])
        assert sorted(order[lo:hi]) == sorted(want_order[lo:hi]), s
    fused = int(seg[25] - seg[17]) == 0 and int(seg[17] - seg[9]) > 0     # (AUTO: both sides of a seed in one launch of the group kernel — 250 bp reads — whose lists put the seeds with an N in EITHER query in front)
    for side, base, qf in ((0, 9, "lqlen"), (1, 17, "rqlen")):
        hn = _gen.query_has_n(tasks, arena, side)
        if fused:
            hn = hn | _gen.query_has_n(tasks, arena, 1 - side)
        for c in range(4):
            idx = order[seg[base + c]:seg[base + c + 1]]
            idx = idx[idx != 0xffffffff]        # (AUTO, a chunk that does not fill the machine: the 8-bit seeds with an N in a query are on the general kernel's list — bsw_binparams.nsplit — and the lists' tails unused)
            key = (~hn[idx]).astype(np.int64) * 1000 - tasks[qf][idx].astype(np.int64)  # queries with an N first, each part longest first
            assert (np.diff(key) >= 0).all()
            if side == 0 and len(idx):
                # inside a query length the left sides go by h0 bucket (8 over the chunk's h0 range): a wave's seeds open their
                # ranges at the same pace.  Checked as: h0 never falls by more than a bucket's width inside one (N, length) run
                h0 = tasks["h0"][idx].astype(np.int64)
                lane_l = order[seg[9]:seg[13]]
                lane_h0 = tasks["h0"][lane_l[lane_l != 0xffffffff]].astype(np.int64)
                width = -(-(int(lane_h0.max()) - int(lane_h0.min()) + 1) // 8)
                same = np.diff(key) == 0
                assert (np.diff(h0)[same] > -width - 1).all()
                if kernel != 1 and len(idx) > 5000:
                    assert (np.diff(h0)[same] < 0).mean() < 0.5            # ... and it is sorted for real, not by accident of the bound
        assert hn.any() and not hn.all()


def test_codes_above_four_are_n(host, oracle, ctx):
    """The pack kernel stores every code > 4 as N (4), like bsw_pack_bases."""
    rng = np.random.default_rng(8)
    seeds = _gen.random_seeds(rng, 600, qmax=200, nrate=0.01)
    tasks, arena = host.make_tasks(seeds)
    dirty = arena.copy()
    pos = rng.random(arena.size) < 0.01
    dirty[pos] = rng.choice([5, 6, 7, 8, 100, 255], pos.sum()).astype(np.uint8)
    clean = np.minimum(dirty, 4)
    p = host.default_params()
    keep, keep2 = np.zeros(arena.size + 8, np.uint8), np.zeros(arena.size + 8, np.uint8)
    t_dirty = rebase(tasks, arena, keep, dirty)
    t_clean = rebase(tasks, arena, keep2, clean)
    got = ctx.extend_pairs(p, t_dirty)
    want = oracle.pair_batch(p, t_clean, nthreads=4)
    assert_same(got, want)


@pytest.mark.parametrize("kernel", [0, 2])
def test_registered_arena_is_dma_direct(host, oracle, kernel):
    """Sequences and results in bsw_host_alloc memory: no gather, no copy-out; results identical."""
    n = 30000
    tasks, arena = host.synth_tasks(n, seed=41, **MIXED)
    p = host.default_params()
    want = oracle.pair_batch(p, tasks, nthreads=8)
    ha = host.HostArena(arena.size + 64)
    ho = host.HostArena(n * host.RESULT.itemsize)
    try:
        t2 = rebase(tasks, arena, ha.u8)
        out = ho.view(host.RESULT, n)
        out[:] = 0
        with host.BswContext(device=0, kernel=kernel, streams=3, chunk_tasks=7000, pack_threads=1) as c:
            got = c.extend_pairs(p, t2, out=out)
            assert_same(got, want, tasks)
            # unregistered output, registered input, and the other way round
            assert_same(c.extend_pairs(p, t2), want, tasks)
            out[:] = 0
            assert_same(c.extend_pairs(p, tasks, out=out), want, tasks)
    finally:
        ha.free()
        ho.free()


def test_host_register_existing_buffer(host, oracle, ctx):
    n = 5000
    tasks, arena = host.synth_tasks(n, seed=43, **MIXED)
    p = host.default_params()
    want = oracle.pair_batch(p, tasks, nthreads=8)
    assert host.host_register(arena) == 0
    try:
        assert_same(ctx.extend_pairs(p, tasks), want, tasks)
    finally:
        assert host.host_unregister(arena) == 0
    assert host.host_unregister(arena) != 0           # not registered any more
    assert_same(ctx.extend_pairs(p, tasks), want, tasks)


def test_one_context_several_devices(host, oracle):
    """bsw_config.devices[]: chunk k -> devices[k mod n].  The box has one GPU, so the same ordinal is listed
    twice (two independent slot sets); the batch manager logic is the one an 8-GPU node runs."""
    n = 50000
    tasks, arena = host.synth_tasks(n, seed=45, **MIXED)
    p = host.default_params()
    want = oracle.pair_batch(p, tasks, nthreads=8)
    with host.BswContext(devices=[0, 0], kernel=host.KERNEL_LANE, streams=2, chunk_tasks=3000) as c:
        assert_same(c.extend_pairs(p, tasks), want, tasks)
        assert_same(c.extend_pairs(p, tasks[:2999]), want[:2999])
    with host.BswContext(devices=[0, 0, 0], streams=1, chunk_tasks=4096) as c:
        assert_same(c.extend_pairs(p, tasks), want, tasks)
    with pytest.raises(host.BswError):
        host.BswContext(devices=[0, 99])


@pytest.mark.parametrize("kernel", [1, 2])
def test_host_supplied_band_limits(host, oracle, kernel):
    """bsw_task.wlim_l/r (the RTL's H5/H6) replace the library's gap-limit formula, in both kernels."""
    n = 6000
    tasks, arena = host.synth_tasks(n, seed=47, **dict(MIXED, indel_rate=0.03))
    rng = np.random.default_rng(2)
    tasks["wlim_l"] = rng.choice([0, 1, 2, 5, 17, 300], n)
    tasks["wlim_r"] = rng.choice([0, 1, 3, 8, 40, 1000], n)
    p = host.default_params()
    want = oracle.pair_batch(p, tasks, nthreads=8)
    t0 = tasks.copy()
    t0["wlim_l"] = 0
    t0["wlim_r"] = 0
    assert oracle.pair_batch(p, t0, nthreads=8).tobytes() != want.tobytes()      # the limits do bind somewhere
    with host.BswContext(device=0, kernel=kernel) as c:
        assert_same(c.extend_pairs(p, tasks), want, tasks)


def test_wire_format_carries_h5_h6(host, oracle, ctx):
    """A header whose max_ins/max_del differ from the library's formula must be honoured."""
    tasks, arena = host.synth_tasks(500, seed=49, **dict(MIXED, indel_rate=0.03, n_rate=0.0))
    rng = np.random.default_rng(3)
    tasks["wlim_l"] = rng.choice([1, 2, 4, 9], len(tasks))
    tasks["wlim_r"] = rng.choice([1, 3, 6, 12], len(tasks))
    p = host.default_params(zdrop=0)
    words, n = host.refbatch_encode(p, tasks)
    assert n == len(tasks)
    p2, t2, seqbuf = host.refbatch_decode(words)
    assert (t2["wlim_l"] == tasks["wlim_l"]).all() and (t2["wlim_r"] == tasks["wlim_r"]).all()
    out, nres = ctx.refbatch_run(words, variant=0, zdrop=0)
    got = host.refbatch_decode_results(out, n)
    want = oracle.pair_batch(p, tasks)
    t0 = tasks.copy()
    t0["wlim_l"] = 0
    t0["wlim_r"] = 0
    assert oracle.pair_batch(p, t0).tobytes() != want.tobytes()
    for f in FIELDS:
        assert (got[f] == want[f]).all(), f


def test_wire_format_nonpositive_h5_h6_reads_as_one(host, oracle, ctx):
    """A malformed header (max_ins/max_del <= 0, never written by bwa) must give the SAME band through bsw_refbatch_run as
    through bsw_refbatch_decode + bsw_submit: both read it as 1 (the wire path used to apply band 0)."""
    tasks, arena = host.synth_tasks(300, seed=53, **dict(MIXED, indel_rate=0.03, n_rate=0.0))
    tasks["wlim_l"] = 3
    tasks["wlim_r"] = 5
    p = host.default_params(zdrop=0)
    words, n = host.refbatch_encode(p, tasks)
    assert n == len(tasks)
    for i in range(n):                               # H5 / H6 of every other task: zero, or a negative max_ins
        if i % 2 == 0:
            words[8 + 8 * i + 5] = 0
            words[8 + 8 * i + 6] = np.uint32(0x0004fffd)      # max_del 4, max_ins -3
    p2, t2, seqbuf = host.refbatch_decode(words)
    assert (t2["wlim_l"][0::2] == 1).all() and (t2["wlim_r"][0::2] == 1).all() and (t2["wlim_l"][1::2] == 3).all()
    out, nres = ctx.refbatch_run(words, variant=0, zdrop=0)
    got = host.refbatch_decode_results(out, n)
    want = oracle.pair_batch(p2, t2)
    via_submit = ctx.extend_pairs(p2, t2)
    assert via_submit.tobytes() == want.tobytes()
    for f in FIELDS:
        assert (got[f] == want[f]).all(), f


@pytest.mark.parametrize("kernel", [0, 2])
def test_wire_batches_in_flight(host, oracle, kernel):
    """bsw_refbatch_submit/wait: many 256 KiB task batches queued, unpacked on the GPU, run as one device batch."""
    p = host.default_params(zdrop=0)
    tasks, arena = host.synth_tasks(40000, seed=51, **dict(MIXED, n_rate=0.002))
    ins, outs, counts, lo = [], [], [], 0
    while lo < len(tasks) and len(ins) < 40:
        words, n = host.refbatch_encode(p, tasks[lo:lo + 819])
        assert n > 0
        ins.append(words)
        outs.append(np.full(host.REFBATCH_OUT_WORDS, 0xdeadbeef, dtype=np.uint32))
        counts.append((lo, n))
        lo += n
    empty = np.zeros(host.REFBATCH_IN_WORDS, dtype=np.uint32)          # a batch with no tasks in the middle
    empty[0], empty[1] = ins[0][0], ins[0][1]
    ins.insert(3, empty)
    outs.insert(3, np.full(host.REFBATCH_OUT_WORDS, 0xdeadbeef, dtype=np.uint32))
    counts.insert(3, (0, 0))
    want = oracle.pair_batch(p, tasks[:lo], nthreads=8)
    with host.BswContext(device=0, kernel=kernel) as c:
        for a, b in zip(ins, outs):
            c.refbatch_submit(a, b)
        assert c.refbatch_wait(variant=0, zdrop=0) == len(ins)
        assert c.refbatch_wait() == 0
    for (l0, n), o in zip(counts, outs):
        got = host.refbatch_decode_results(o, n)
        for f in FIELDS:
            assert (got[f] == want[l0:l0 + n][f]).all(), (l0, f)
        assert (o[5 * n:] == 0).all()


def test_wire_pipeline_errors_and_header_changes(host, oracle):
    """The wire path runs groups of 16 batches, four in flight: a malformed batch in a late group is an error that leaves
    nothing in flight (the context keeps working), and a change of the scoring header in mid-queue starts a new group."""
    pa, pb = host.default_params(zdrop=0), host.default_params(zdrop=0, o_del=4, o_ins=4, e_del=2, e_ins=2, w=40)
    tasks, arena = host.synth_tasks(70 * 819, seed=61, **MIXED)
    ins, outs, meta, lo = [], [], [], 0
    k = 0
    while lo < len(tasks) and len(ins) < 70:
        p = pb if 20 <= k < 45 else pa                                   # header changes twice, not on group boundaries
        words, n = host.refbatch_encode(p, tasks[lo:lo + 819])
        ins.append(words); outs.append(np.zeros(host.REFBATCH_OUT_WORDS, np.uint32)); meta.append((lo, n, p))
        lo += n; k += 1
    with host.BswContext(device=0) as c:
        for a, b in zip(ins, outs):
            c.refbatch_submit(a, b)
        assert c.refbatch_wait(0, 0) == len(ins)
        for (l0, n, p), o in zip(meta, outs):
            want = oracle.pair_batch(p, tasks[l0:l0 + n], nthreads=8)
            got = host.refbatch_decode_results(o, n)
            for f in FIELDS:
                assert (got[f] == want[f]).all(), (l0, f)
        bad = ins[50].copy()
        bad[8 + 2] = 0x7fffffff                                         # first task's data position far outside the batch
        for i, (a, b) in enumerate(zip(ins, outs)):
            c.refbatch_submit(bad if i == 50 else a, b)
        with pytest.raises(host.BswError):
            c.refbatch_wait(0, 0)
        assert c.refbatch_wait() == 0                                   # queue dropped, nothing in flight
        for a, b in zip(ins[:5], outs[:5]):
            b[:] = 0
            c.refbatch_submit(a, b)
        assert c.refbatch_wait(0, 0) == 5
        l0, n, p = meta[2]
        got = host.refbatch_decode_results(outs[2], n)
        want = oracle.pair_batch(p, tasks[l0:l0 + n], nthreads=8)
        assert (got["score"] == want["score"]).all()


def test_scalar_abi_from_many_threads(host, oracle):
    """bwa's -t worker threads call ksw_extend2 concurrently: calls are coalesced into device batches."""
    L = host.lib()
    m = host.bwa_matrix()
    nthr, per = 16, 300
    rng = np.random.default_rng(4)
    cases = []
    for k in range(nthr * per):
        ql, tl = int(rng.integers(1, 150)), int(rng.integers(0, 260))
        t = rng.integers(0, 4, tl).astype(np.uint8)
        q = _gen.mutate(rng, t, ql, 0.04, 0.02)
        cases.append((q, t, int(rng.choice([10, 100, 200])), int(rng.choice([0, 5])), int(rng.choice([0, 100])), int(rng.integers(1, 80))))
    res = [None] * len(cases)
    c0, t0 = host.scalar_stats()

    def work(tid):
        for k in range(tid * per, (tid + 1) * per):
            q, t, w, eb, zd, h0 = cases[k]
            outs = [C.c_int(0) for _ in range(5)]
            sc = L.ksw_extend2(len(q), q.ctypes.data, len(t), t.ctypes.data if len(t) else None, 5, m.ctypes.data, 6, 1, 6, 1, w, eb, zd, h0,
                               *[C.addressof(o) for o in outs])
            res[k] = (sc,) + tuple(o.value for o in outs)

    th = [threading.Thread(target=work, args=(i,)) for i in range(nthr)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    for k, (q, t, w, eb, zd, h0) in enumerate(cases):
        r = oracle.extend2(q, t, m, 6, 1, 6, 1, w, eb, zd, h0)
        assert res[k] == (r["score"], r["qle"], r["tle"], r["gtle"], r["gscore"], r["max_off"]), k
    c1, t1 = host.scalar_stats()
    assert c1 - c0 == len(cases) and 0 < t1 - t0 <= len(cases)
    # neutral results outside bwa's domain (h0 <= 0, qlen == 0): no GPU round trip, outputs always written
    outs = [C.c_int(7) for _ in range(5)]
    q = np.array([0, 1, 2, 3], np.uint8)
    assert L.ksw_extend2(4, q.ctypes.data, 4, q.ctypes.data, 5, m.ctypes.data, 6, 1, 6, 1, 100, 5, 100, 0, *[C.addressof(o) for o in outs]) == 0
    assert [o.value for o in outs] == [0, 0, 0, -1, 0]
    assert L.ksw_extend2(0, None, 4, q.ctypes.data, 5, m.ctypes.data, 6, 1, 6, 1, 100, 5, 100, 9, None, None, None, None, None) == 9


def test_all_three_scalar_entry_points_share_round_trips(host, oracle):
    """ksw_extend2, ksw_align2 and ksw_global2 called concurrently from 12 threads (bwa's -t workers in mem_chain2aln, mate
    rescue and CIGAR generation): one queue, coalesced device trips, every result equal to the oracle's (the
    alignment calls used to be one serialised round trip each)."""
    L = host.lib()
    m = host.bwa_matrix()

    class KSWR(C.Structure):
        _fields_ = [(f, C.c_int) for f in ("score", "te", "qe", "score2", "te2", "tb", "qb")]
    L.ksw_align2.restype = KSWR
    L.ksw_align2.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p]
    nthr, per = 12, 60
    rng = np.random.default_rng(14)
    cases = []
    for k in range(nthr * per):
        tl = int(rng.integers(60, 500))
        t = rng.integers(0, 4, tl).astype(np.uint8)
        ql = int(rng.integers(20, 150))
        q = _gen.mutate(rng, t[int(rng.integers(0, 30)):], ql, 0.04, 0.02)
        cases.append((k % 3, q, t, int(rng.integers(1, 60))))
    res = [None] * len(cases)
    c0, t0 = host.scalar_stats()
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]

    def work(tid):
        for k in range(tid * per, (tid + 1) * per):
            kind, q, t, h0 = cases[k]
            if kind == 0:
                outs = [C.c_int(0) for _ in range(5)]
                sc = L.ksw_extend2(len(q), q.ctypes.data, len(t), t.ctypes.data, 5, m.ctypes.data, 6, 1, 6, 1, 100, 5, 100, h0, *[C.addressof(o) for o in outs])
                res[k] = (sc,) + tuple(o.value for o in outs)
            elif kind == 1:
                r = L.ksw_align2(len(q), q.ctypes.data, len(t), t.ctypes.data, 5, m.ctypes.data, 6, 1, 6, 1, 0x40000 | 0x80000 | 19, None)
                res[k] = tuple(getattr(r, f) for f in ("score", "te", "qe", "score2", "te2", "tb", "qb"))
            else:
                nc, cg = C.c_int(0), C.POINTER(C.c_uint32)()
                tt = t[:len(q) + 8]
                sc = L.ksw_global2(len(q), q.ctypes.data, len(tt), tt.ctypes.data, 5, m.ctypes.data, 6, 1, 6, 1, 30, C.addressof(nc), C.addressof(cg))
                res[k] = (sc, [cg[i] for i in range(nc.value)])
                libc.free(cg)

    th = [threading.Thread(target=work, args=(i,)) for i in range(nthr)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    for k, (kind, q, t, h0) in enumerate(cases):
        if kind == 0:
            r = oracle.extend2(q, t, m, 6, 1, 6, 1, 100, 5, 100, h0)
            assert res[k] == (r["score"], r["qle"], r["tle"], r["gtle"], r["gscore"], r["max_off"]), k
        elif kind == 1:
            w = oracle.align2(q, t, m, 6, 1, 6, 1, 0x40000 | 0x80000 | 19)
            assert res[k] == tuple(w[f] for f in ("score", "te", "qe", "score2", "te2", "tb", "qb")), k
        else:
            tt = t[:len(q) + 8]
            g = oracle.global2(q, tt, m, 6, 1, 6, 1, 30)
            assert res[k][0] == g["score"] and [(c & 0xf, c >> 4) for c in res[k][1]] == g["cigar"], k
    c1, t1 = host.scalar_stats()
    assert c1 - c0 == len(cases) and 0 < t1 - t0 < len(cases)          # every call counted; fewer trips than calls: they were shared


def test_busy_context_refuses_other_entry_points(host):
    tasks, arena = host.synth_tasks(200000, seed=53)
    p = host.default_params()
    with host.BswContext(device=0) as c:
        out = c.submit(p, tasks)
        with pytest.raises(host.BswError) as ei:
            c.upload(p, tasks[:10])
        assert ei.value.code == -6
        more = [c.submit(p, tasks[:10]) for _ in range(host.MAX_INFLIGHT - 1)]        # ABI 6: a context keeps MAX_INFLIGHT submits going
        with pytest.raises(host.BswError) as ei:
            c.submit(p, tasks[:10])
        assert ei.value.code == -6 and c.inflight() == host.MAX_INFLIGHT
        c.wait()
        assert c.inflight() == 0
        assert (out["tag"] == np.arange(len(tasks), dtype=np.uint32)).all()
        assert all((m["tag"] == np.arange(10, dtype=np.uint32)).all() for m in more)


def test_watchdog_marks_the_context_dead(host):
    """A wait for the GPU that exceeds timeout_ms fails with BSW_E_HIP and every later call fails fast
    (the batch here is simply larger than a 1 ms deadline allows; nothing hangs).  Sequences and results sit in
    registered memory so that the whole round trip is asynchronous and the wait is where the time goes."""
    n = 600000
    ha = host.HostArena(host.synth_arena_bound(n) + 4096)
    ho = host.HostArena(n * host.RESULT.itemsize)
    tasks, _ = host.synth_tasks(n, arena=ha.u8, seed=55)
    p = host.default_params()
    c = host.BswContext(device=0, timeout_ms=1, streams=1, chunk_tasks=n)
    with pytest.raises(host.BswError) as ei:
        c.extend_pairs(p, tasks, out=ho.view(host.RESULT, n))
    assert ei.value.code == -4 and "timeout" in str(ei.value)
    with pytest.raises(host.BswError) as ei:
        c.upload(p, tasks[:10])
    assert ei.value.code == -4
    c.close()
    with host.BswContext(device=0) as c2:                   # the device itself is fine
        assert len(c2.extend_pairs(p, tasks[:100])) == 100
    import time
    time.sleep(0.05)                                        # let the abandoned batch drain before its buffers go
    ha.free()
    ho.free()


@pytest.mark.parametrize("devflag", [["--gpus", "1"], ["--devices", "0,0"], ["--gpus", "1", "--pageable"]])
def test_bsw_bench_cli(host, oracle, devflag):
    """The C host over the C ABI: its result checksum must equal the one computed from the oracle's result batch."""
    import json
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tools", "bsw-bench")
    n = 50000
    r = subprocess.run([exe, "-n", str(n), "-b", "16384"] + devflag, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    j = json.loads(r.stdout.strip().splitlines()[-1])
    tasks, arena = host.synth_tasks(n, seed=1, read_len=150, seed_len_min=19, seed_len_max=60, seed_at_start=0,
                                    sub_rate=0.01, indel_rate=0.001, junk_frac=0.05, n_rate=0.0, a=1, w=100, o=6, e=1)
    want = oracle.pair_batch(host.default_params(), tasks, nthreads=8)
    s, M = 0, (1 << 64) - 1
    for sc, ts, qb, re in zip(want["score"].tolist(), want["truesc"].tolist(), want["qb"].tolist(), want["re"].tolist()):
        s = (s * 1315423911 + (sc & 0xffffffff) + ((ts & 0xffffffff) << 20) + ((qb * 7) & 0xffffffff) + ((re * 13) & 0xffffffff)) & M
    assert j["seeds"] == n and j["result_checksum"] == "%016x" % s
    assert j["cells"] == int(want["left"]["cells"].astype(np.int64).sum() + want["right"]["cells"].astype(np.int64).sum())
    # the same seeds handed over 4-bit packed, and with bwa's -O del,ins / -E del,ins syntax on the variant-M recurrence
    rp = subprocess.run([exe, "-n", str(n), "-b", "16384", "--packed"] + devflag, capture_output=True, text=True, timeout=300)
    assert rp.returncode == 0, rp.stderr
    jp = json.loads(rp.stdout.strip().splitlines()[-1])
    assert jp["packed_input"] == 1 and jp["result_checksum"] == j["result_checksum"] and jp["cells"] == j["cells"]
    rg = subprocess.run([exe, "-n", "20000", "-O", "6,4", "-E", "1,2", "--variant=M", "--packed"] + devflag, capture_output=True, text=True, timeout=300)
    assert rg.returncode == 0, rg.stderr
    jg = json.loads(rg.stdout.strip().splitlines()[-1])
    tg, _ = host.synth_tasks(20000, seed=1, read_len=150, seed_len_min=19, seed_len_max=60, seed_at_start=0,
                             sub_rate=0.01, indel_rate=0.001, junk_frac=0.05, n_rate=0.0, a=1, w=100, o=6, e=1)
    wg = oracle.pair_batch(host.default_params(o_del=6, o_ins=4, e_del=1, e_ins=2, variant=1), tg, nthreads=8)
    assert jg["cells"] == int(wg["left"]["cells"].astype(np.int64).sum() + wg["right"]["cells"].astype(np.int64).sum())


def test_deep_wire_queue_runs_in_groups_of_64(host, oracle):
    """A queue of >= 96 task batches is cut into groups of >= 64 (52 k seeds: the two-seeds-per-lane kernels, two groups in
    flight, enqueued side by side) instead of 16 (the general kernel): the same result batches either way, batch by batch,
    against the oracle — including a last group that takes the queue's remainder (100 batches: 64 + 33 + 3) and a header change
    inside the queue.  The result batches are written on the device: everything behind a batch's last record is zero."""
    pa, pb = host.default_params(zdrop=0), host.default_params(zdrop=0, o_del=5, o_ins=5, w=60)
    tasks, arena = host.synth_tasks(100 * 819, seed=77, **MIXED)
    ins, outs, meta, lo, k = [], [], [], 0, 0
    while lo < len(tasks) and len(ins) < 100:
        p = pb if k >= 97 else pa
        words, n = host.refbatch_encode(p, tasks[lo:lo + 819])
        ins.append(words); outs.append(np.full(host.REFBATCH_OUT_WORDS, 0xdeadbeef, np.uint32)); meta.append((lo, n, p))
        lo += n; k += 1
    wa, wb = oracle.pair_batch_avx2(pa, tasks[:lo], nthreads=8), oracle.pair_batch(pb, tasks[:lo], nthreads=8)
    with host.BswContext(device=0) as c:
        for a, b in zip(ins, outs):
            c.refbatch_submit(a, b)
        assert c.refbatch_wait(0, 0) == len(ins)
        for (l0, n, p), o in zip(meta, outs):
            want = (wb if p is pb else wa)[l0:l0 + n]
            got = host.refbatch_decode_results(o, n)
            for f in FIELDS:
                assert (got[f] == want[f]).all(), (l0, f)
            assert (o[5 * n:] == 0).all()
        # the big groups of a deep queue are parsed and enqueued side by side on host threads: a malformed batch in the SECOND
        # group (a helper thread's) is the call's error, nothing stays in flight, and the context keeps working
        bad = ins[80].copy()
        bad[8 + 2] = 0x7fffffff
        for i, (a, b) in enumerate(zip(ins, outs)):
            c.refbatch_submit(bad if i == 80 else a, b)
        with pytest.raises(host.BswError, match="malformed task batch"):
            c.refbatch_wait(0, 0)
        assert c.refbatch_wait() == 0
        for o in outs:
            o[:] = 0xdeadbeef
        for a, b in zip(ins, outs):
            c.refbatch_submit(a, b)
        assert c.refbatch_wait(0, 0) == len(ins)
        for (l0, n, p), o in zip(meta, outs):
            got = host.refbatch_decode_results(o, n)
            assert (got["score"] == (wb if p is pb else wa)[l0:l0 + n]["score"]).all() and (o[5 * n:] == 0).all()


def test_registered_task_batches_are_dmad_where_they_are(host, oracle):
    """Task batches in memory from bsw_host_alloc skip the copy into pinned staging: one DMA per run of batches that lie back
    to back (here: 20 contiguous ones, then 10 with gaps between them), mixed with a pageable batch in the same group (the whole
    group then takes the staging path).  Result batches in such memory are DMA'd where they are as well (a group whose result
    batches are not ALL registered goes through pinned staging).  Same result batches as the oracle either way."""
    p = host.default_params(zdrop=0)
    tasks, arena = host.synth_tasks(31 * 819, seed=88, **MIXED)
    W = host.REFBATCH_IN_WORDS
    ar = host.HostArena(45 * W * 4)
    view = ar.view(np.uint32, 45 * W).reshape(45, W)
    slots = list(range(20)) + list(range(21, 41, 2))                    # 20 back to back, 10 every other slot
    # the result batches of the first 25 likewise (back to back, then every other slot): written where they are; the rest pageable
    OW = host.REFBATCH_OUT_WORDS
    oar = host.HostArena(45 * OW * 4)
    oview = oar.view(np.uint32, 45 * OW).reshape(45, OW)
    ins, outs, meta, lo = [], [], [], 0
    for k, sl in enumerate(slots):
        words, n = host.refbatch_encode(p, tasks[lo:lo + 819])
        view[sl] = words
        ins.append(view[sl]); outs.append(oview[sl] if k < 25 else np.zeros(OW, np.uint32)); meta.append((lo, n)); lo += n
    want = oracle.pair_batch(p, tasks[:lo], nthreads=8)
    with host.BswContext(device=0) as c:
        for rep in range(2):
            extra = []
            if rep == 1:                                                 # a pageable batch joins the last group
                words, n = host.refbatch_encode(p, tasks[lo:lo + 819])
                extra = [(words, np.zeros(host.REFBATCH_OUT_WORDS, np.uint32), (lo, n))]
            for o in outs:
                o[:] = 0xdeadbeef
            for a, b in zip(ins, outs):
                c.refbatch_submit(a, b)
            for a, b, _ in extra:
                c.refbatch_submit(a, b)
            assert c.refbatch_wait(0, 0) == len(ins) + len(extra)
            for (l0, n), o in zip(meta, outs):
                got = host.refbatch_decode_results(o, n)
                for f in FIELDS:
                    assert (got[f] == want[l0:l0 + n][f]).all(), (rep, l0, f)
                assert (o[5 * n:] == 0).all()
            for a, b, (l0, n) in extra:
                got = host.refbatch_decode_results(b, n)
                w2 = oracle.pair_batch(p, tasks[l0:l0 + n], nthreads=8)
                assert (got["score"] == w2["score"]).all() and (got["truesc"] == w2["truesc"]).all()
    ar.free(); oar.free()
