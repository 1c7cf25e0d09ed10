"""bsw_align_batch / ksw_align2 on the GPU (bwa's striped local alignment of mate rescue) against the
oracle's literal emulation: every output field, both modes, all flag combinations, the saturation and length edges."""
import ctypes as C

import numpy as np
import pytest

import _gen

pytestmark = pytest.mark.gpu
XB, XSTOP, XSUBO, XSTART = 0x10000, 0x20000, 0x40000, 0x80000
FIELDS = ("score", "te", "qe", "score2", "te2", "tb", "qb")


def make(host, pairs, xtras):
    at = np.zeros(len(pairs), dtype=host.ATASK)
    keep = []
    for i, ((q, t), x) in enumerate(zip(pairs, xtras)):
        q, t = np.ascontiguousarray(q, np.uint8), np.ascontiguousarray(t, np.uint8)
        keep.append((q, t))
        at[i]["query"], at[i]["target"], at[i]["qlen"], at[i]["tlen"], at[i]["xtra"] = q.ctypes.data, t.ctypes.data, len(q), len(t), x
    return at, keep


def check(host, oracle, ctx, p, at):
    got = ctx.align_batch(p, at)
    want, _ = oracle.align2_batch(p["mat"][0], int(p["o_del"][0]), int(p["e_del"][0]), int(p["o_ins"][0]), int(p["e_ins"][0]), at, nthreads=8)
    for k, f in enumerate(FIELDS):
        bad = np.nonzero(got[f] != want[:, k])[0]
        assert bad.size == 0, "%s: task %s (qlen %s tlen %s xtra %s) got %s want %s" % (
            f, bad[:4], at["qlen"][bad[:4]], at["tlen"][bad[:4]], [hex(x) for x in at["xtra"][bad[:4]]], got[f][bad[:4]], want[bad[:4], k])
    return got


@pytest.fixture(scope="module")
def ctx(host):
    with host.BswContext(device=0) as c:
        yield c


def rescue_like(rng, n, qmax, tmax, junk=0.2):
    pairs = []
    for _ in range(n):
        tl = int(rng.integers(1, tmax))
        t = rng.integers(0, 4, tl).astype(np.uint8)
        ql = int(rng.integers(1, qmax))
        if rng.random() < junk:
            q = rng.integers(0, 5, ql).astype(np.uint8)
        else:
            a = int(rng.integers(0, max(1, tl - ql)))
            q = _gen.mutate(rng, t[a:a + ql], ql, 0.05, 0.02)
        if rng.random() < 0.1:
            q[rng.integers(0, len(q))] = 4
        if len(q):
            pairs.append((q, t))
    return pairs


@pytest.mark.parametrize("mode", ["byte", "word"])
def test_mate_rescue_shapes(host, oracle, ctx, mode):
    rng = np.random.default_rng(1 if mode == "byte" else 2)
    p = host.default_params()
    pairs = rescue_like(rng, 3000, 151, 700)
    base = XB if mode == "byte" else 0
    xt = [base | int(rng.choice([0, XSTART, XSUBO | 19, XSUBO | XSTART | 19, XSUBO | XSTART | 30, XSTOP | 25])) for _ in pairs]
    got = check(host, oracle, ctx, p, make(host, pairs, xt)[0])
    assert (got["score"] > 50).sum() > 400 and (got["tb"] >= 0).sum() > 300 and (got["score2"] >= 0).sum() > 10


def test_every_query_length_and_vector_count(host, oracle, ctx):
    """Query lengths 1..256 hit every slen of both modes (the striping changes with it) and both kernel classes per mode."""
    rng = np.random.default_rng(3)
    p = host.default_params()
    pairs, xt = [], []
    for ql in range(1, 257):
        t = rng.integers(0, 4, int(rng.integers(ql, 2 * ql + 40))).astype(np.uint8)
        a = int(rng.integers(0, len(t) - ql + 1))
        q = _gen.mutate(rng, t[a:a + ql], ql, 0.03, 0.02)
        if len(q) == 0:
            continue
        for x in (XB | XSUBO | XSTART | 19, XSUBO | XSTART | 19):
            pairs.append((q, t)); xt.append(x)
    check(host, oracle, ctx, p, make(host, pairs, xt)[0])


def test_saturation_limits_and_degenerate_inputs(host, oracle, ctx):
    rng = np.random.default_rng(4)
    p = host.default_params()
    q = rng.integers(0, 4, 256).astype(np.uint8)
    t = np.concatenate([rng.integers(0, 4, 20), q, rng.integers(0, 4, 20)]).astype(np.uint8)
    pairs = [(q, t), (q, t), (q[:250], t), (q[:251], t), (np.zeros(30, np.uint8), np.full(90, 3, np.uint8)),
             (q[:40], np.zeros(0, np.uint8)), (np.full(60, 4, np.uint8), t), (q[:1], t[:1])]
    xt = [XB | XSTART, XSTART, XB | XSTART, XB | XSTART, XB | XSTART, XB, XSTART | XSUBO | 1, XB | XSTART]
    got = check(host, oracle, ctx, p, make(host, pairs, xt)[0])
    assert got["score"][0] == 255 and got["score"][1] == 256 and got["score"][2] == 250      # 8-bit run saturates, 16-bit does not
    with pytest.raises(host.BswError):
        ctx.align_batch(p, make(host, [(np.zeros(1025, np.uint8), t)], [0])[0])              # beyond BSW_ALIGN_MAX_QLEN: an error, not a fallback


@pytest.mark.parametrize("mode", ["byte", "word"])
def test_long_queries(host, oracle, ctx, mode):
    """Reads longer than 256 bases (2x300 mate rescue): the 512- and 1024-column classes of both modes, every
    vector count boundary, scores far beyond the 8-bit range in word mode and saturating in byte mode."""
    rng = np.random.default_rng(11 if mode == "byte" else 12)
    p = host.default_params()
    pairs, xt = [], []
    base = XB if mode == "byte" else 0
    for ql in list(range(257, 1025, 37)) + [256, 257, 300, 511, 512, 513, 1000, 1023, 1024]:
        t = rng.integers(0, 4, int(rng.integers(ql, ql + 600))).astype(np.uint8)
        a = int(rng.integers(0, len(t) - ql + 1))
        q = _gen.mutate(rng, t[a:a + ql], ql, 0.06 if rng.random() < 0.7 else 0.3, 0.02)
        if rng.random() < 0.2:
            q = rng.integers(0, 5, ql).astype(np.uint8)
        for x in (XSUBO | XSTART | 19, XSTART, XSUBO | 40, 0):
            pairs.append((q, t)); xt.append(base | x)
    got = check(host, oracle, ctx, p, make(host, pairs, xt)[0])
    if mode == "word":
        assert got["score"].max() > 300 and (got["tb"] >= 0).sum() > 20          # far beyond the 8-bit range
    else:
        assert (got["score"] == 255).sum() > 10


@pytest.mark.parametrize("seed", range(4))
def test_scoring_and_gap_penalties(host, oracle, ctx, seed):
    rng = np.random.default_rng(20 + seed)
    a, b, nn = [(1, 4, -1), (2, 3, -2), (1, 1, 0), (3, 6, -1)][seed]
    p = host.default_params(o_del=int(rng.integers(0, 12)), e_del=int(rng.integers(1, 5)), o_ins=int(rng.integers(0, 12)), e_ins=int(rng.integers(1, 5)))
    p["mat"][0] = host.bwa_matrix(a, b, nn)
    pairs = rescue_like(rng, 1500, 200 if a < 3 else 80, 500)
    xt = [int(rng.choice([XB, 0])) | XSUBO | XSTART | int(rng.integers(5, 40)) for _ in pairs]
    for i, (q, t) in enumerate(pairs):                                  # keep the 8-bit runs inside their range or let them saturate: both are checked
        pass
    check(host, oracle, ctx, p, make(host, pairs, xt)[0])


def test_drop_in_ksw_align2(host, oracle):
    """The scalar ABI with bwa's signature (kswr_t by value)."""
    L = host.lib()

    class KSWR(C.Structure):
        _fields_ = [(f, C.c_int) for f in FIELDS]
    L.ksw_align2.restype = KSWR
    L.ksw_align2.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p]
    L.ksw_align.restype = KSWR
    L.ksw_align.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p] + [C.c_int] * 3 + [C.c_void_p]
    rng = np.random.default_rng(6)
    m = host.bwa_matrix()
    for _ in range(30):
        t = rng.integers(0, 4, int(rng.integers(50, 400))).astype(np.uint8)
        q = _gen.mutate(rng, t[10:10 + int(rng.integers(20, 150))], 100, 0.04, 0.02)
        x = XB | XSUBO | XSTART | 19
        r = L.ksw_align2(len(q), q.ctypes.data, len(t), t.ctypes.data, 5, m.ctypes.data, 6, 1, 6, 1, x, None)
        w = oracle.align2(q, t, m, 6, 1, 6, 1, x)
        assert [getattr(r, f) for f in FIELDS] == [w[f] for f in FIELDS]
        r = L.ksw_align(len(q), q.ctypes.data, len(t), t.ctypes.data, 5, m.ctypes.data, 5, 2, x, None)
        w = oracle.align2(q, t, m, 5, 2, 5, 2, x)
        assert [getattr(r, f) for f in FIELDS] == [w[f] for f in FIELDS]


def test_one_over_limit_call_does_not_fail_the_other_threads(host, oracle):
    """Concurrent ksw_align2 / ksw_global2 calls share a device batch; a batch API rejects the whole batch on its first bad
    task.  One thread's over-limit query (2 000 bases > BSW_ALIGN_MAX_QLEN) must come back as ITS failure only."""
    import threading
    L = host.lib()

    class KSWR(C.Structure):
        _fields_ = [(f, C.c_int) for f in FIELDS]
    L.ksw_align2.restype = KSWR
    L.ksw_align2.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p]
    L.ksw_global2.restype = C.c_int
    L.ksw_global2.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(66)
    m = host.bwa_matrix()
    x = XB | XSUBO | XSTART | 19
    nthr, rounds = 12, 6
    work = []
    for k in range(nthr):
        t = rng.integers(0, 4, 300).astype(np.uint8)
        q = _gen.mutate(rng, t[20:170], 120, 0.03, 0.01)
        work.append((q, t))
    bigq = rng.integers(0, 4, 2000).astype(np.uint8)
    bigt = rng.integers(0, 4, 2500).astype(np.uint8)
    got = [[None] * rounds for _ in range(nthr)]
    gsc = [[None] * rounds for _ in range(nthr)]
    bad = [None] * rounds
    gate = threading.Barrier(nthr + 1)

    def worker(k):
        q, t = work[k]
        for r in range(rounds):
            gate.wait()
            res = L.ksw_align2(len(q), q.ctypes.data, len(t), t.ctypes.data, 5, m.ctypes.data, 6, 1, 6, 1, x, None)
            got[k][r] = [getattr(res, f) for f in FIELDS]
            gsc[k][r] = L.ksw_global2(len(q), q.ctypes.data, len(q), t.ctypes.data + 20, 5, m.ctypes.data, 6, 1, 6, 1, 30, None, None)

    def offender():
        for r in range(rounds):
            gate.wait()
            res = L.ksw_align2(len(bigq), bigq.ctypes.data, len(bigt), bigt.ctypes.data, 5, m.ctypes.data, 6, 1, 6, 1, x, None)
            bad[r] = res.score

    th = [threading.Thread(target=worker, args=(k,)) for k in range(nthr)] + [threading.Thread(target=offender)]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join()
    assert all(b == -1 for b in bad)                                    # the over-limit call fails ...
    for k in range(nthr):                                               # ... and nobody else does
        q, t = work[k]
        w = oracle.align2(q, t, m, 6, 1, 6, 1, x)
        g = oracle.global2(q, t[20:20 + len(q)], m, 6, 1, 6, 1, 30, want_cigar=False)["score"]
        for r in range(rounds):
            assert got[k][r] == [w[f] for f in FIELDS], (k, r)
            assert gsc[k][r] == g, (k, r)
