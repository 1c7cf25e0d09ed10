"""GPU parity: the HIP path (through the C ABI) vs the CPU oracle — bit-exact on every field of
every result record, including the per-side raw ksw_extend2 outputs and the exact cell counts."""
import ctypes as C

import numpy as np
import pytest

import _gen
import _golden

pytestmark = pytest.mark.gpu

FIELDS = ["tag", "qb", "qe", "rb", "re", "score", "truesc", "w"]
EXTF = ["score", "qle", "tle", "gtle", "gscore", "max_off", "aw", "cells"]


def assert_same(got, want, tasks=None):
    if got.tobytes() == want.tobytes():
        return
    for f in FIELDS:
        bad = np.nonzero(got[f] != want[f])[0]
        assert bad.size == 0, "field %s differs at %s: got %s want %s" % (f, bad[:5], got[f][bad[:5]], want[f][bad[:5]])
    for side in ("left", "right"):
        for f in EXTF:
            bad = np.nonzero(got[side][f] != want[side][f])[0]
            assert bad.size == 0, "%s.%s differs at %s: got %s want %s (task %s)" % (
                side, f, bad[:5], got[side][f][bad[:5]], want[side][f][bad[:5]],
                None if tasks is None else tasks[bad[:1]])
    raise AssertionError("byte difference outside named fields")


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("zdrop", [0, 100])
def test_mixed_seeds(host, oracle, ctx, variant, zdrop):
    p = host.default_params(variant=variant, zdrop=zdrop)
    tasks, arena = host.synth_tasks(4096, seed=11 + variant, seed_len_min=19, seed_len_max=60, seed_at_start=0,
                                    sub_rate=0.02, indel_rate=0.01, junk_frac=0.15, n_rate=0.002)
    got = ctx.extend_pairs(p, tasks)
    want = oracle.pair_batch(p, tasks, nthreads=8)
    assert_same(got, want, tasks)


def test_single_bin_150bp(host, oracle, ctx):
    p = host.default_params()
    tasks, arena = host.synth_tasks(8192, seed=3)          # BASELINE config[1] shape: qlen=131, tlen=257
    assert (tasks["rqlen"] == 131).all() and (tasks["rtlen"] == 257).all()
    got = ctx.extend_pairs(p, tasks)
    want = oracle.pair_batch(p, tasks, nthreads=8)
    assert_same(got, want, tasks)


def test_250bp_w500(host, oracle, ctx):
    p = host.default_params(w=500)
    tasks, arena = host.synth_tasks(3000, seed=8, read_len=250, seed_len_min=19, seed_len_max=40, seed_at_start=0,
                                    sub_rate=0.04, indel_rate=0.01, junk_frac=0.05, n_rate=0.001, w=500)
    got = ctx.extend_pairs(p, tasks)
    want = oracle.pair_batch(p, tasks, nthreads=8)
    assert_same(got, want, tasks)


# every kernel class boundary: columns per lane C in {1,2,3,4,8,16} <=> qlen+1 <= 64*C of the register kernels, then the two
# LDS-row classes of bsw_long_kernel.hip (2 048 and 8 192 columns: BSW_MAX_QLEN = 8 191)
@pytest.mark.parametrize("qlen", [1, 2, 62, 63, 64, 126, 127, 128, 190, 191, 192, 254, 255, 256, 510, 511, 512, 1022, 1023,
                                  1024, 1500, 2047, 2048, 4096, 8191])
def test_class_boundaries(host, oracle, ctx, qlen):
    rng = np.random.default_rng(qlen)
    seeds = []
    for k in range(24 if qlen < 1024 else 6):
        tl = int(qlen * 1.3) + 5
        t = rng.integers(0, 4, tl).astype(np.uint8)
        q = _gen.mutate(rng, t, qlen, 0.03, 0.01 if k % 2 else 0.0)
        s = {"rq": q, "rt": t, "h0": int(rng.integers(5, 50))}
        if k % 3 == 0:
            s["lq"], s["lt"] = q[::-1].copy(), t[::-1].copy()
        seeds.append(s)
    tasks, arena = host.make_tasks(seeds)
    for variant in (0, 1):
        p = host.default_params(variant=variant)
        assert_same(ctx.extend_pairs(p, tasks), oracle.pair_batch(p, tasks, nthreads=4), tasks)


@pytest.mark.parametrize("over", [
    dict(w=1), dict(w=5, zdrop=0), dict(w=37, zdrop=7), dict(zdrop=1), dict(max_band_try=1), dict(max_band_try=3, w=8),
    dict(o_del=0, e_del=1, o_ins=0, e_ins=1), dict(o_del=5, e_del=2, o_ins=7, e_ins=1), dict(o_del=11, e_del=3, o_ins=2, e_ins=4),
    dict(pen_clip5=0, pen_clip3=0), dict(pen_clip5=20, pen_clip3=1),
])
@pytest.mark.parametrize("variant", [0, 1])
def test_parameter_space(host, oracle, ctx, over, variant):
    rng = np.random.default_rng(len(str(over)) + variant)
    seeds = _gen.random_seeds(rng, 700, qmax=150, indel=0.04, junk=0.15, nrate=0.003)
    tasks, arena = host.make_tasks(seeds)
    p = host.default_params(variant=variant, **over)
    assert_same(ctx.extend_pairs(p, tasks), oracle.pair_batch(p, tasks, nthreads=8), tasks)


@pytest.mark.parametrize("variant", [0, 1])
def test_general_scoring_matrix(host, oracle, ctx, variant):
    rng = np.random.default_rng(123 + variant)
    seeds = _gen.random_seeds(rng, 600, qmax=120, nrate=0.02)
    tasks, arena = host.make_tasks(seeds)
    for it in range(3):
        p = host.default_params(variant=variant)
        p["mat"][0] = rng.integers(-7, 5, 25).astype(np.int8)
        assert_same(ctx.extend_pairs(p, tasks), oracle.pair_batch(p, tasks, nthreads=8), tasks)
    p = host.default_params(variant=variant)
    p["mat"][0] = host.bwa_matrix(a=2, b=3, n=-2)
    assert_same(ctx.extend_pairs(p, tasks), oracle.pair_batch(p, tasks, nthreads=8), tasks)


def test_edge_shapes(host, oracle, ctx):
    z = np.zeros(0, np.uint8)
    a40 = (np.arange(40) % 4).astype(np.uint8)
    seeds = [
        dict(rq=a40, rt=z, h0=9),                                   # tlen == 0
        dict(rq=a40[:1], rt=a40[:1], h0=1),                         # qlen == 1
        dict(lq=a40, lt=a40, h0=30),                                # left side only
        dict(lq=a40, lt=z, rq=a40, rt=z, h0=3),                     # both targets empty
        dict(rq=a40, rt=np.tile(a40, 100), h0=10),                  # tlen 4000 >> qlen + band: beg runs past end
        dict(rq=np.full(100, 4, np.uint8), rt=np.full(150, 4, np.uint8), h0=50),   # all N
        dict(rq=a40, rt=(a40 + 1) % 4, h0=1),                       # dies in row 0
        dict(rq=np.tile(a40, 20), rt=np.tile(a40, 25), h0=127),     # 800 x 1000 perfect match, class 16
        dict(lq=a40, lt=a40, rq=a40, rt=a40, h0=19, init_score=59), # prev == score on the left -> no retry
        dict(rq=a40, rt=a40, h0=1000),                              # large h0: row 0 band fully alive
    ]
    tasks, arena = host.make_tasks(seeds)
    for variant in (0, 1):
        for zd in (0, 100):
            p = host.default_params(variant=variant, zdrop=zd)
            assert_same(ctx.extend_pairs(p, tasks), oracle.pair_batch(p, tasks), tasks)


def test_long_targets_cross_1024_chunks(host, oracle, ctx):
    rng = np.random.default_rng(4242)
    seeds = []
    for k in range(40):
        ql = int(rng.integers(100, 900))
        t = rng.integers(0, 4, int(rng.integers(1000, 5000))).astype(np.uint8)
        seeds.append(dict(rq=_gen.mutate(rng, t, ql, 0.02, 0.003), rt=t, h0=int(rng.integers(10, 80))))
    tasks, arena = host.make_tasks(seeds)
    p = host.default_params(w=500, zdrop=0)
    assert_same(ctx.extend_pairs(p, tasks), oracle.pair_batch(p, tasks, nthreads=8), tasks)


def test_band_retry_is_exercised(host, oracle, ctx):
    rng = np.random.default_rng(7)
    seeds = []
    for k in range(300):
        q = rng.integers(0, 4, 140).astype(np.uint8)
        gap = int(rng.integers(60, 130))
        if k % 2:
            t = np.concatenate([q[:50], rng.integers(0, 4, gap).astype(np.uint8), q[50:], rng.integers(0, 4, 40).astype(np.uint8)])
            seeds.append(dict(rq=q, rt=t, h0=60))
        else:
            qq = np.concatenate([q[:50], rng.integers(0, 4, gap // 2).astype(np.uint8), q[50:]])
            seeds.append(dict(lq=qq, lt=np.concatenate([q, rng.integers(0, 4, 100).astype(np.uint8)]), h0=60))
    tasks, arena = host.make_tasks(seeds)
    p = host.default_params(w=60, zdrop=0)
    want = oracle.pair_batch(p, tasks, nthreads=8)
    assert (want["w"] == 120).sum() > 20, "workload does not trigger MAX_BAND_TRY"
    assert_same(ctx.extend_pairs(p, tasks), want, tasks)


@pytest.mark.parametrize("name", _golden.names())
def test_golden_fixtures(host, ctx, name):
    tasks, arena, cases = _golden.load(host, name)
    for pname, (params, expect) in cases.items():
        assert_same(ctx.extend_pairs(params, tasks), expect, tasks)


def test_streaming_submit_chunks(host, oracle):
    tasks, arena = host.synth_tasks(10000, seed=17, seed_len_min=19, seed_len_max=60, seed_at_start=0, junk_frac=0.1)
    p = host.default_params()
    want = oracle.pair_batch(p, tasks, nthreads=8)
    for streams, chunk in ((1, 999), (3, 1024), (2, 100000)):
        with host.BswContext(device=0, streams=streams, chunk_tasks=chunk, pack_threads=3) as c:
            assert_same(c.extend_pairs(p, tasks), want, tasks)
            assert_same(c.extend_pairs(p, tasks[:1]), want[:1])
            assert len(c.extend_pairs(p, tasks[:0])) == 0


def test_device_resident_rerun_is_idempotent(host, oracle, ctx):
    tasks, arena = host.synth_tasks(5000, seed=23)
    p = host.default_params()
    b = ctx.upload(p, tasks)
    ctx.run(b)
    r1 = ctx.download(b)
    ctx.run(b)
    ctx.run(b)
    r2 = ctx.download(b)
    ms = ctx.last_run_ms()
    hist = ctx.run_history()
    b.free()
    assert r1.tobytes() == r2.tobytes()
    assert ms > 0 and len(hist) >= 3
    assert_same(r1, oracle.pair_batch(p, tasks, nthreads=8), tasks)


def test_scalar_abi_ksw_extend(host, oracle):
    L = host.lib()
    rng = np.random.default_rng(1)
    m = host.bwa_matrix()
    for variant in (0, 1):
        L.bsw_set_default_variant(variant)
        for it in range(12):
            ql, tl = int(rng.integers(1, 200)), int(rng.integers(0, 300))
            t = rng.integers(0, 4, tl).astype(np.uint8)
            q = _gen.mutate(rng, t, ql, 0.05, 0.02)
            outs = [C.c_int(0) for _ in range(5)]
            w, eb, zd, h0 = int(rng.choice([5, 50, 100])), int(rng.integers(0, 8)), int(rng.choice([0, 100])), int(rng.integers(1, 60))
            if it % 2:
                sc = L.ksw_extend2(ql, q.ctypes.data, tl, t.ctypes.data, 5, m.ctypes.data, 5, 2, 7, 1, w, eb, zd, h0,
                                   *[C.addressof(o) for o in outs])
                ref = oracle.extend2(q, t, m, 5, 2, 7, 1, w, eb, zd, h0, variant=variant)
            else:
                sc = L.ksw_extend(ql, q.ctypes.data, tl, t.ctypes.data, 5, m.ctypes.data, 6, 1, w, eb, zd, h0,
                                  *[C.addressof(o) for o in outs])
                ref = oracle.extend2(q, t, m, 6, 1, 6, 1, w, eb, zd, h0, variant=variant)
            got = dict(score=sc, qle=outs[0].value, tle=outs[1].value, gtle=outs[2].value, gscore=outs[3].value, max_off=outs[4].value)
            ref.pop("cells")
            assert got == ref, (variant, it)
    L.bsw_set_default_variant(0)
    # NULL out-pointers are allowed (bwa passes NULL for values it does not need)
    q = np.array([0, 1, 2, 3], np.uint8)
    assert L.ksw_extend2(4, q.ctypes.data, 4, q.ctypes.data, 5, m.ctypes.data, 6, 1, 6, 1, 100, 5, 100, 10, None, None, None, None, None) == 14


def test_extend_batch_mixed_bands(host, oracle, ctx):
    rng = np.random.default_rng(3)
    n = 500
    et = np.zeros(n, dtype=host.EXT_TASK)
    keep = []
    for i in range(n):
        ql, tl = int(rng.integers(1, 180)), int(rng.integers(0, 260))
        t = rng.integers(0, 4, tl).astype(np.uint8)
        q = _gen.mutate(rng, t, ql, 0.04, 0.02)
        keep.append((q, t))
        et[i]["query"], et[i]["target"] = q.ctypes.data, t.ctypes.data if tl else 0
        et[i]["qlen"], et[i]["tlen"] = ql, tl
        et[i]["w"], et[i]["end_bonus"], et[i]["h0"] = int(rng.choice([3, 20, 100, 200])), int(rng.choice([0, 5])), int(rng.integers(1, 70))
    for variant in (0, 1):
        p = host.default_params(variant=variant)
        got = ctx.extend_batch(p, et)
        want = oracle.ext_batch(p, et, nthreads=4)
        for f in EXTF:
            assert (got[f] == want[f]).all(), f


def test_reference_wire_format_end_to_end(host, oracle, ctx):
    tasks, arena = host.synth_tasks(700, seed=31, seed_at_start=0, seed_len_min=19, seed_len_max=60, indel_rate=0.01, junk_frac=0.1)
    p = host.default_params()
    words, n = host.refbatch_encode(p, tasks)
    assert n == 700
    out, nres = ctx.refbatch_run(words, variant=0, zdrop=0)
    assert nres == n
    got = host.refbatch_decode_results(out, n)
    p0 = host.default_params(zdrop=0)
    want = oracle.pair_batch(p0, tasks)
    for f in FIELDS:
        assert (got[f] == want[f]).all(), f
    assert (out[5 * n:] == 0).all()


def test_limits_are_errors_not_fallbacks(host, ctx):
    p = host.default_params()
    t, a = host.make_tasks([dict(rq=np.zeros(8192, np.uint8), rt=np.zeros(10, np.uint8), h0=5)])       # BSW_MAX_QLEN = 8191
    with pytest.raises(host.BswError) as ei:
        ctx.extend_pairs(p, t)
    assert ei.value.code == -3
    t, a = host.make_tasks([dict(rq=np.zeros(10, np.uint8), rt=np.zeros(10, np.uint8), h0=0)])
    with pytest.raises(host.BswError) as ei:
        ctx.extend_pairs(p, t)
    assert ei.value.code == -2
    bad = host.default_params(e_ins=0)
    t, a = host.make_tasks([dict(rq=np.zeros(10, np.uint8), rt=np.zeros(10, np.uint8), h0=3)])
    with pytest.raises(host.BswError):
        ctx.extend_pairs(bad, t)


def test_full_size_properties(host, oracle, ctx):
    """BASELINE configs[1] size (1M seeds): properties that do not need the oracle at full size,
    plus an oracle check on a strided sample."""
    n = 1_000_000
    tasks, arena = host.synth_tasks(n, seed=1000)
    p = host.default_params()
    b = ctx.upload(p, tasks)
    ctx.run(b)
    res = ctx.download(b)
    b.free()
    r = res["right"]
    assert (res["tag"] == np.arange(n, dtype=np.uint32)).all()
    assert (r["score"] >= 19).all() and (r["score"] <= 19 + 131).all()
    assert (r["qle"] >= 0).all() and (r["qle"] <= 131).all() and (r["tle"] <= 257).all() and (r["gtle"] <= 257).all()
    assert (r["cells"] <= 131 * 257 * 2).all() and (r["cells"] > 0).all()
    assert (res["left"]["cells"] == 0).all()
    assert (res["score"] == r["score"]).all()
    # permutation invariance: a shuffled batch gives the permuted result batch
    perm = np.random.default_rng(0).permutation(200_000)
    sub = tasks[:200_000][perm].copy()
    got = ctx.extend_pairs(p, sub)
    assert got.tobytes() == res[:200_000][perm].tobytes()
    # strided oracle sample
    idx = np.arange(0, n, 97)
    want = oracle.pair_batch(p, tasks[idx].copy(), nthreads=8)
    assert_same(res[idx], want, tasks[idx])


def test_config2_10m_mixed_bins_streaming(host, oracle):
    """BASELINE configs[2] size: 10M 150 bp PE seeds, mixed bins, through the streaming batch manager
    (host buffers in, host buffers out); oracle parity on a strided sample + size-independent properties."""
    n = 10_000_000
    tasks, arena = host.synth_tasks(n, seed=77, seed_len_min=19, seed_len_max=60, seed_at_start=0,
                                    junk_frac=0.05, n_rate=0.0005)
    p = host.default_params()
    with host.BswContext(device=0, streams=3, chunk_tasks=65536, pack_threads=8) as c:
        res = c.extend_pairs(p, tasks)
    assert (res["tag"] == np.arange(n, dtype=np.uint32)).all()
    assert (res["score"] >= tasks["h0"]).all() and (res["score"] <= 150).all()
    assert (res["qb"] >= 0).all() and (res["qb"] <= tasks["qbeg"]).all() and (res["qe"] <= tasks["rqlen"]).all()
    assert (res["left"]["cells"][tasks["lqlen"] == 0] == 0).all()
    assert ((res["w"] == 100) | (res["w"] == 200)).all()
    idx = np.arange(0, n, 997)
    want = oracle.pair_batch(p, tasks[idx].copy(), nthreads=16)
    assert_same(res[idx], want, tasks[idx])


def test_extend_batch_large_goes_through_lane_bins(host, oracle, ctx):
    """Batched plain ksw_extend2 above the BSW_KERNEL_AUTO threshold: served by the lane-per-extension kernel."""
    n = host.LANE_AUTO_MIN + 4000
    tasks, arena = host.synth_tasks(n, seed=61, seed_len_min=19, seed_len_max=50, seed_at_start=1, indel_rate=0.01)
    et = np.zeros(n, dtype=host.EXT_TASK)
    et["query"], et["target"], et["qlen"], et["tlen"] = tasks["rquery"], tasks["rtarget"], tasks["rqlen"], tasks["rtlen"]
    et["w"], et["end_bonus"], et["h0"] = 100, 5, tasks["h0"]
    for variant in (0, 1):
        p = host.default_params(variant=variant)
        got = ctx.extend_batch(p, et)
        want = oracle.ext_batch(p, et, nthreads=8)
        for f in EXTF:
            assert (got[f] == want[f]).all(), f


def test_long_queries_through_every_entry_point(host, oracle, ctx):
    """Queries beyond the register kernels' 1 023 bases (bwa's ksw_extend2 has no limit): a mixed batch of
    short and long seeds through bsw_submit, a resident batch, the packed path and the drop-in scalar ksw_extend2; one base
    more than BSW_MAX_QLEN is BSW_E_LIMIT, never a fallback."""
    rng = np.random.default_rng(8191)
    seeds = []
    for k, qlen in enumerate([30, 1100, 131, 3000, 2047, 75, 5000, 1024, 8191, 400]):
        tl = int(qlen * 1.2) + 20
        t = rng.integers(0, 4, tl).astype(np.uint8)
        q = _gen.mutate(rng, t, qlen, 0.04, 0.005)
        s = {"rq": q, "rt": t, "h0": int(rng.integers(19, 60))}
        if k % 2:
            lq = _gen.mutate(rng, t[::-1].copy(), max(qlen // 3, 1), 0.03, 0.0)
            s["lq"], s["lt"] = lq, t[::-1].copy()
        seeds.append(s)
    tasks, arena = host.make_tasks(seeds)
    for variant, w in ((0, 100), (1, 37), (0, 2000)):
        p = host.default_params(variant=variant, w=w)
        want = oracle.pair_batch(p, tasks, nthreads=4)
        assert_same(ctx.extend_pairs(p, tasks), want, tasks)
        b = ctx.upload(p, tasks); ctx.run(b); got = ctx.download(b); b.free()
        assert_same(got, want, tasks)
        pt, pa = host.pack_tasks(tasks)
        assert_same(ctx.extend_pairs_packed(p, pt), want, tasks)
    # the drop-in scalar entry point on a 3 000-base query
    q, t = host.task_seq(tasks, 3, "rquery", "rqlen"), host.task_seq(tasks, 3, "rtarget", "rtlen")
    m = host.bwa_matrix()
    outs = [C.c_int(0) for _ in range(5)]
    r = host.lib().ksw_extend2(len(q), q.ctypes.data, len(t), t.ctypes.data, 5, m.ctypes.data, 6, 1, 6, 1, 100, 5, 100, 40,
                               *[C.byref(o) for o in outs])
    w = oracle.extend2(q, t, m, 6, 1, 6, 1, 100, 5, 100, 40)
    assert (r, outs[0].value, outs[1].value, outs[2].value, outs[3].value, outs[4].value) == (w["score"], w["qle"], w["tle"], w["gtle"], w["gscore"], w["max_off"])
    # one base too many
    big = rng.integers(0, 4, 8192).astype(np.uint8)
    bad, _a = host.make_tasks([{"rq": big, "rt": big, "h0": 20}])
    with pytest.raises(host.BswError) as ei:
        ctx.extend_pairs(host.default_params(), bad)
    assert ei.value.code == -3
