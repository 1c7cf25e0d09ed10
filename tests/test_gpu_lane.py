"""Lane-per-task kernel (inter-task SIMD bins) vs the CPU oracle, forced with BSW_KERNEL_LANE so even
small batches go through it; the wave-per-task kernel handles what the lane bins cannot take
(N, long queries, general matrices) and every seed whose first band try is not final (redo list)."""
import numpy as np
import pytest

import _gen
import _golden
from test_gpu_parity import assert_same

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lctx(host):
    c = host.BswContext(device=0, kernel=host.KERNEL_LANE)
    yield c
    c.close()


@pytest.fixture(scope="module")
def wctx(host):
    c = host.BswContext(device=0, kernel=host.KERNEL_WAVE)
    yield c
    c.close()


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("zdrop", [0, 100])
def test_single_bin(host, oracle, lctx, variant, zdrop):
    p = host.default_params(variant=variant, zdrop=zdrop)
    tasks, arena = host.synth_tasks(6000, seed=40 + variant)
    assert_same(lctx.extend_pairs(p, tasks), oracle.pair_batch(p, tasks, nthreads=8), tasks)


@pytest.mark.parametrize("variant", [0, 1])
def test_mixed_bins_both_sides(host, oracle, lctx, wctx, variant):
    p = host.default_params(variant=variant)
    tasks, arena = host.synth_tasks(9000, seed=50 + variant, seed_len_min=19, seed_len_max=60, seed_at_start=0,
                                    sub_rate=0.02, indel_rate=0.01, junk_frac=0.2, n_rate=0.001)
    want = oracle.pair_batch(p, tasks, nthreads=8)
    assert_same(lctx.extend_pairs(p, tasks), want, tasks)
    assert_same(wctx.extend_pairs(p, tasks), want, tasks)


@pytest.mark.parametrize("over", [
    dict(w=1), dict(w=5, zdrop=0), dict(w=37, zdrop=7), dict(zdrop=1), dict(max_band_try=1), dict(max_band_try=3, w=8),
    dict(o_del=0, e_del=1, o_ins=0, e_ins=1), dict(o_del=5, e_del=2, o_ins=7, e_ins=1), dict(o_del=11, e_del=3, o_ins=2, e_ins=4),
    dict(pen_clip5=0, pen_clip3=0), dict(pen_clip5=20, pen_clip3=1),
])
@pytest.mark.parametrize("variant", [0, 1])
def test_parameter_space(host, oracle, lctx, over, variant):
    rng = np.random.default_rng(1000 + len(str(over)) + variant)
    seeds = _gen.random_seeds(rng, 900, qmax=134, indel=0.04, junk=0.15, nrate=0.0005)
    tasks, arena = host.make_tasks(seeds)
    p = host.default_params(variant=variant, **over)
    assert_same(lctx.extend_pairs(p, tasks), oracle.pair_batch(p, tasks, nthreads=8), tasks)


@pytest.mark.parametrize("ab", [(1, 4), (2, 3), (3, 0), (5, 9)])
def test_match_mismatch_scores(host, oracle, lctx, ab):
    rng = np.random.default_rng(sum(ab))
    seeds = _gen.random_seeds(rng, 700, qmax=120, indel=0.03, h0max=200)
    tasks, arena = host.make_tasks(seeds)
    for variant in (0, 1):
        p = host.default_params(variant=variant)
        p["mat"][0] = host.bwa_matrix(a=ab[0], b=ab[1], n=-1)
        assert_same(lctx.extend_pairs(p, tasks), oracle.pair_batch(p, tasks, nthreads=8), tasks)


# lane classes hold qlen+1 <= 136 / 232 columns; the narrow values exercise block skipping
@pytest.mark.parametrize("qlen", [1, 2, 7, 8, 9, 31, 32, 33, 63, 64, 70, 71, 72, 102, 103, 104, 134, 135, 136])
def test_lane_class_boundaries(host, oracle, lctx, qlen):
    rng = np.random.default_rng(500 + qlen)
    seeds = []
    for k in range(200):
        tl = int(qlen * 1.5) + int(rng.integers(0, 20))
        t = rng.integers(0, 4, tl).astype(np.uint8)
        q = _gen.mutate(rng, t, qlen, 0.03, 0.01 if k % 2 else 0.0)
        s = {"rq": q, "rt": t, "h0": int(rng.integers(1, 60))}
        if k % 3 == 0:
            ql2 = int(rng.integers(1, qlen + 1))
            s["lq"], s["lt"] = q[:ql2][::-1].copy(), t[::-1].copy()
        seeds.append(s)
    tasks, arena = host.make_tasks(seeds)
    for variant in (0, 1):
        p = host.default_params(variant=variant)
        assert_same(lctx.extend_pairs(p, tasks), oracle.pair_batch(p, tasks, nthreads=4), tasks)


# the 8-bit 232-column lane class (250 bp reads): h0 + qlen*a must stay <= 255
@pytest.mark.parametrize("qlen", [136, 137, 180, 229, 230, 231])
def test_lane_class_250bp(host, oracle, lctx, qlen):
    rng = np.random.default_rng(900 + qlen)
    seeds = []
    for k in range(160):
        tl = int(qlen * 1.4) + int(rng.integers(0, 30))
        t = rng.integers(0, 4, tl).astype(np.uint8)
        q = _gen.mutate(rng, t, qlen, 0.04, 0.01 if k % 2 else 0.0)
        seeds.append({"rq": q, "rt": t, "h0": int(rng.integers(1, 256 - qlen))})
    tasks, arena = host.make_tasks(seeds)
    for variant in (0, 1):
        for w in (100, 500):
            p = host.default_params(variant=variant, w=w)
            assert_same(lctx.extend_pairs(p, tasks), oracle.pair_batch(p, tasks, nthreads=4), tasks)


@pytest.mark.parametrize("ab", [(1, 4), (2, 3), (1, 1), (3, 7)])
def test_top_of_the_8_bit_score_range(host, oracle, lctx, ab):
    """Perfect matches that end exactly at, just below and just above 255 - b.  The two-seeds-per-lane kernel forms
    H + a + b in 8 bits, so the batch manager may give it a seed only while h0 + qlen*a + b <= 255; anything above goes
    to a wider class — and every one of them must still be exact."""
    a, b = ab
    rng = np.random.default_rng(a * 10 + b)
    seeds = []
    for k in range(2000):
        lq = int(rng.integers(0, 60))
        rq = int(rng.integers(1, min(135, (255 - 1) // a - lq)))
        h0 = 255 - b - (lq + rq) * a + int(rng.integers(-2, b + 1))        # top in [255-b-2, 255]
        if h0 < 1:
            continue
        qL, qR = rng.integers(0, 4, lq).astype(np.uint8), rng.integers(0, 4, rq).astype(np.uint8)
        seeds.append(dict(lq=qL, lt=np.concatenate([qL, rng.integers(0, 4, 20).astype(np.uint8)]),
                          rq=qR, rt=np.concatenate([qR, rng.integers(0, 4, 20).astype(np.uint8)]), h0=h0))
    tasks, arena = host.make_tasks(seeds)
    p = host.default_params()
    p["mat"][0] = host.bwa_matrix(a, b, -1)
    want = oracle.pair_batch(p, tasks, nthreads=4)
    assert want["score"].max() == 255
    assert_same(lctx.extend_pairs(p, tasks), want, tasks)


def test_250bp_w500_workload(host, oracle, lctx):
    p = host.default_params(w=500)
    tasks, arena = host.synth_tasks(5000, seed=18, read_len=250, seed_len_min=19, seed_len_max=40, seed_at_start=0,
                                    sub_rate=0.04, indel_rate=0.01, junk_frac=0.05, n_rate=0.0005, w=500)
    assert_same(lctx.extend_pairs(p, tasks), oracle.pair_batch(p, tasks, nthreads=8), tasks)


@pytest.mark.parametrize("nscore", [-1, -3, 0, 1])
def test_n_bases_in_lane_bins(host, oracle, lctx, nscore):
    """N in query and/or target: every pair involving an N scores mat[4][.] (K6)."""
    rng = np.random.default_rng(70 + nscore)
    seeds = _gen.random_seeds(rng, 800, qmax=134, indel=0.02, junk=0.1, nrate=0.03)
    seeds += [dict(rq=np.full(100, 4, np.uint8), rt=np.full(150, 4, np.uint8), h0=50),          # all N
              dict(rq=np.full(60, 4, np.uint8), rt=rng.integers(0, 4, 90).astype(np.uint8), h0=70),
              dict(lq=rng.integers(0, 4, 60).astype(np.uint8), lt=np.full(90, 4, np.uint8), h0=70)]
    tasks, arena = host.make_tasks(seeds)
    for variant in (0, 1):
        p = host.default_params(variant=variant)
        p["mat"][0] = host.bwa_matrix(a=1, b=4, n=nscore)
        assert_same(lctx.extend_pairs(p, tasks), oracle.pair_batch(p, tasks, nthreads=8), tasks)


def test_edge_shapes(host, oracle, lctx):
    z = np.zeros(0, np.uint8)
    a40 = (np.arange(40) % 4).astype(np.uint8)
    seeds = [
        dict(rq=a40, rt=z, h0=9), dict(rq=a40[:1], rt=a40[:1], h0=1), dict(lq=a40, lt=a40, h0=30),
        dict(lq=a40, lt=z, rq=a40, rt=z, h0=3), dict(rq=a40, rt=np.tile(a40, 100), h0=10),
        dict(rq=a40, rt=(a40 + 1) % 4, h0=1), dict(lq=a40, lt=a40, rq=a40, rt=a40, h0=19, init_score=59),
        dict(rq=a40, rt=a40, h0=1000), dict(rq=np.tile(a40, 3), rt=np.tile(a40, 30), h0=60000),
        dict(rq=np.tile(a40, 3), rt=np.tile(a40, 4), h0=64000),     # beyond the 16-bit bins -> wave kernel
    ] * 7
    tasks, arena = host.make_tasks(seeds)
    for variant in (0, 1):
        for zd in (0, 100):
            p = host.default_params(variant=variant, zdrop=zd)
            assert_same(lctx.extend_pairs(p, tasks), oracle.pair_batch(p, tasks), tasks)


def _retry_seeds():
    rng = np.random.default_rng(7)
    seeds = []
    for k in range(600):
        q = rng.integers(0, 4, 90).astype(np.uint8)
        tail = rng.integers(0, 4, 60).astype(np.uint8)
        if k % 3 == 0:
            seeds.append(dict(rq=q, rt=np.concatenate([q, tail]), h0=40))                       # no retry
            continue
        gap = int(rng.integers(38, 50))                                                         # >= 3/4 of w=50, inside the band
        if k % 2:
            t = np.concatenate([q[:45], rng.integers(0, 4, gap).astype(np.uint8), q[45:], tail])      # long deletion, right side
            seeds.append(dict(rq=q, rt=t, h0=60))
        else:
            qq = np.concatenate([q[:40], rng.integers(0, 4, gap).astype(np.uint8), q[40:]])[:134]     # long insertion, left side
            seeds.append(dict(lq=qq, lt=np.concatenate([q, tail]), rq=q[:30], rt=q[:50], h0=60))
    return seeds


def test_band_retry_goes_through_redo_list(host, oracle, lctx):
    tasks, arena = host.make_tasks(_retry_seeds())
    p = host.default_params(w=50, zdrop=0)
    want = oracle.pair_batch(p, tasks, nthreads=8)
    assert (want["w"] == 100).sum() > 30 and (want["w"] == 50).sum() > 30
    assert_same(lctx.extend_pairs(p, tasks), want, tasks)


@pytest.mark.parametrize("name", _golden.names())
def test_golden_fixtures(host, lctx, name):
    tasks, arena, cases = _golden.load(host, name)
    for pname, (params, expect) in cases.items():
        assert_same(lctx.extend_pairs(params, tasks), expect, tasks)


def test_ragged_last_wave_and_tiny_batches(host, oracle, lctx):
    p = host.default_params()
    tasks, arena = host.synth_tasks(1000, seed=77, seed_len_min=19, seed_len_max=50, seed_at_start=0)
    want = oracle.pair_batch(p, tasks, nthreads=4)
    for n in (1, 2, 63, 64, 65, 255, 256, 257, 1000):
        assert_same(lctx.extend_pairs(p, tasks[:n]), want[:n], tasks[:n])


NARROW_SNIPPET = r"""
import sys, numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import __graft_entry__ as g
host, orc = g.load_package().host, g.load_oracle()
from test_gpu_parity import assert_same
for variant, gaps in ((0, {}), (1, dict(o_del=5, e_del=2, o_ins=7, e_ins=1))):
    p = host.default_params(variant=variant, **gaps)
    tasks, arena = host.synth_tasks(40000, seed=77 + variant, seed_len_min=19, seed_len_max=60, seed_at_start=0, sub_rate=0.02,
                                    indel_rate=0.01, junk_frac=0.1, n_rate=0.001)
    with host.BswContext(device=0, kernel=host.KERNEL_LANE) as c:
        b = c.upload(p, tasks); c.run(b); c.sync()
        got, launches = c.download(b), b.info()["launches"]
        b.free()
    assert launches == 5, launches              # left + right x (72-column class, 136-column class) + the redo list
    assert_same(got, orc.pair_batch(p, tasks, nthreads=8), tasks)
print("ok")
"""


@pytest.mark.parametrize("fork", ["0", "1"])
def test_narrow_class_beside_the_wide_one(fork):
    """The 72-column class forced on (BSW_NARROW_SHARE=0) in a chunk that also holds wider sides — by default such a chunk
    folds it away — with the classes of a side one after the other and side by side on forked streams (BSW_FORK=1): a right
    side waits for exactly the left launches that hold its seeds.  The switches are read once per process: own process."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BSW_NARROW_SHARE="0", BSW_FORK=fork)
    out = subprocess.run([sys.executable, "-c", NARROW_SNIPPET % dict(root=root)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout[-2000:] + out.stderr[-4000:]


CHAIN_SNIPPET = r"""
import sys, numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import __graft_entry__ as g
host, orc = g.load_package().host, g.load_oracle()
from test_gpu_parity import assert_same
for variant, gaps in ((0, {}), (1, dict(o_del=5, e_del=2, o_ins=7, e_ins=1))):
    p = host.default_params(variant=variant, w=500, **gaps)
    tasks, arena = host.synth_tasks(60000, seed=91 + variant, read_len=250, seed_len_min=19, seed_len_max=40, seed_at_start=0, sub_rate=0.04,
                                    indel_rate=0.01, junk_frac=0.05, n_rate=0.0005, w=500)
    want = orc.pair_batch(p, tasks, nthreads=8)
    with host.BswContext(device=0, kernel=host.KERNEL_LANE) as c:
        for rep in range(3):                        # the flags are reset and raised again by every run
            b = c.upload(p, tasks); c.run(b); c.run(b); c.sync()
            got, launches = c.download(b), b.info()["launches"]
            b.free()
            assert launches == %(launches)d, launches           # left + right x the lane classes in use + the redo list (the last side finishes its seeds: no finalize launch)
            assert_same(got, want, tasks)
        got = c.extend_pairs(p, tasks)              # the synchronous chunk path shares stream 0 and its chain
        assert_same(got, want, tasks)
        assert c.chain_timeouts() == 0              # in normal operation no waiting wave ever reaches its 20 ms deadline
print("ok")
"""


@pytest.mark.parametrize("mode", ["chain", "chain3", "0", "1"])
def test_lane_launches_as_a_chain_on_250_bp_reads(mode):
    """250 bp reads have sides in the 136- and the 232-column class.  By default the chunk's lane launches form a chain on
    four streams — left narrow, left wide, right wide, right narrow — each released by the launch before it: the kernels
    count their started workgroups in a device word, one sleeping wave in front of the next launch polls it until it reads
    the grid size, so the follower takes the ragged end of the wide launch and never a slot it could still use (chain3: with the
    72-column class forced on, six launches); BSW_FORK=0: one stream; BSW_FORK=1: the classes of a side released together."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("BSW_FORK", None)
    if mode in ("0", "1"):
        env["BSW_FORK"] = mode
    if mode == "chain3":
        env["BSW_NARROW_SHARE"] = "0"
    out = subprocess.run([sys.executable, "-c", CHAIN_SNIPPET % dict(root=root, launches=7 if mode == "chain3" else 5)], env=env,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout[-2000:] + out.stderr[-4000:]


WEDGE_SNIPPET = r"""
import sys, time, numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import __graft_entry__ as g
host, orc = g.load_package().host, g.load_oracle()
from test_gpu_parity import assert_same
p = host.default_params(w=500)
tasks, arena = host.synth_tasks(40000, seed=93, read_len=250, seed_len_min=19, seed_len_max=40, seed_at_start=0, sub_rate=0.04,
                                indel_rate=0.01, junk_frac=0.05, n_rate=0.0005, w=500)
want = orc.pair_batch(p, tasks, nthreads=8)
with host.BswContext(device=0, kernel=host.KERNEL_LANE, timeout_ms=60000) as c:
    t0 = time.time()
    for rep in range(2):
        b = c.upload(p, tasks); c.run(b); c.sync()
        got, launches = c.download(b), b.info()["launches"]
        b.free()
        assert launches == 5, launches               # the chain is ON: four lane launches + the redo list
        assert_same(got, want, tasks)
    got = c.extend_pairs(p, tasks)
    assert_same(got, want, tasks)
    print("timeouts", c.chain_timeouts(), "seconds", round(time.time() - t0, 2))
print("ok")
"""


@pytest.mark.parametrize("setting", ["serialize", "blocking", "deadline"])
def test_launch_chain_cannot_wedge(setting):
    """The chain's waiting wave (bsw_wait_count) is BOUNDED: the word it polls is a scheduling hint, the data dependencies are
    stream events, so where something runs kernels one at a time — and may pick the waiter before the launch that raises its
    word — the wave gives up after 20 ms and the follower starts.  Round 4 sniffed five environment names and switched the
    chain off; now the chain stays ON (six launches) under AMD_SERIALIZE_KERNEL=3 and HIP_LAUNCH_BLOCKING=1, and with
    BSW_CHAIN_SELFTEST every wait is made to run into its deadline: results bit-exact, the process finishes, the expired
    waits are counted (reference: the TBB -> PE array -> RBB hand-off is an FSM that cannot wedge, tbb.v:110-123, rbb.v:219-224)."""
    import os, re, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("BSW_FORK", "BSW_CHAIN_SELFTEST", "AMD_SERIALIZE_KERNEL", "HIP_LAUNCH_BLOCKING"):
        env.pop(k, None)
    if setting == "serialize":
        env["AMD_SERIALIZE_KERNEL"] = "3"
    elif setting == "blocking":
        env["HIP_LAUNCH_BLOCKING"] = "1"
    else:
        env["BSW_CHAIN_SELFTEST"] = "1"
    out = subprocess.run([sys.executable, "-c", WEDGE_SNIPPET % dict(root=root)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout[-2000:] + out.stderr[-4000:]
    m = re.search(r"timeouts (\d+) seconds ([0-9.]+)", out.stdout)
    assert m, out.stdout
    if setting == "deadline":
        assert int(m.group(1)) == 6, out.stdout          # two resident runs of a four-link chain: three waits each, all expired
                                                         # (extend_pairs streams through the slot pipeline, which does not chain)
    assert float(m.group(2)) < 30.0, out.stdout          # and a wait that expires costs 20 ms, not a watchdog


def test_mid_sized_chunk_runs_the_group_kernel(host, oracle):
    """Between the general kernels and the lane kernels: a resident batch whose launched sides hold 1.5 M - 5 M query bases runs
    bsw_lane2g_kernel (a seed pair per group of eight lanes) under BSW_KERNEL_AUTO.  45 000 PE seeds, some of them outside the
    8-bit score range (they leave the lane lists for the general kernel in this mode), both variants, separate gap penalties."""
    n = 45_000
    tasks, arena = host.synth_tasks(n, seed=77, seed_len_min=19, seed_len_max=60, seed_at_start=0, sub_rate=0.02, indel_rate=0.01,
                                    junk_frac=0.1, n_rate=0.002)
    tasks["h0"][::53] = 300                                              # 16-bit seeds
    for variant, gaps in ((0, {}), (1, dict(o_del=5, e_del=2, o_ins=7, e_ins=1))):
        p = host.default_params(variant=variant, **gaps)
        order, seg, words = host.plan_batch(p, tasks, kernel=host.KERNEL_AUTO)
        n16 = int((tasks["h0"] == 300).sum())
        assert seg[9] - seg[8] == n - n16 - int(((tasks["lqlen"] == 0) & (tasks["rqlen"] == 0) & (tasks["h0"] != 300)).sum())      # lane seeds = the 8-bit ones
        assert seg[8] - seg[0] >= n16                                     # the 16-bit ones sit in the wave classes
        want = oracle.pair_batch(p, tasks, nthreads=8)
        with host.BswContext(device=0) as c:
            b = c.upload(p, tasks); c.run(b); got = c.download(b)
            assert b.info()["launches"] >= 3                              # left + right group launches, the redo launch (+ wave classes)
            b.free()
            assert_same(got, want, tasks)
            assert_same(c.extend_pairs(p, tasks), want, tasks)            # the same through a submit (one chunk)


@pytest.mark.parametrize("read_len,n", [(150, 30_000), (250, 20_000), (150, 90_000)])
def test_group_kernel_runs_both_sides_of_a_seed_in_one_launch(host, oracle, read_len, n):
    """A mid-sized chunk of two-sided seeds (up to GROUP_FUSE_MAX of them): ONE bsw_lane2g_kernel launch over all left lists — a
    wavefront runs the left sides of its 16 seeds, then their right sides with the scores it just found as h0 — so the chunk
    takes one wave lifetime instead of two.  Seeds without a left side (they sit at query length 0 of the left lists), without
    a right side (finished by the first half), 16-bit seeds (general kernel), Ns, both variants, separate gap penalties, full
    and pair records; 250 bp reads take the four-stripe instantiation.  Past GROUP_FUSE_MAX seeds (the third case) the LANE kernels
    do the same up to LANE_FUSE_MAX (bsw_lane2_kernel's fused instantiation); there the 16-bit seeds keep their own lane class with a
    list per side."""
    spec = dict(read_len=read_len, seed_len_min=19, seed_len_max=60, seed_at_start=0, sub_rate=0.02, indel_rate=0.01, junk_frac=0.1,
                n_rate=0.0005 if n > 50_000 else 0.001 if read_len > 200 else 0.002)      # (the N list is capped by its work — NLIST_WORK_MAX, 1.8 M query bases: a sample of the chunk decides; these stay well below)
    if read_len == 250:
        spec.update(w=500, seed_len_max=20)                               # (h0 + 231 + b <= 255: longer seeds of 250 bp reads are 16-bit seeds)
    ta, a1 = host.synth_tasks(n, seed=91, **spec)
    tb, a2 = host.synth_tasks(n // 8, seed=92, **dict(spec, seed_at_start=1))       # no left side
    tasks = np.concatenate([ta, tb])
    tasks["tag"] = np.arange(len(tasks), dtype=np.uint32)
    noright = np.arange(5, len(ta), 23)
    tasks["rqlen"][noright] = 0
    tasks["rtlen"][noright] = 0
    tasks["h0"][::61] = 300                                              # 16-bit seeds
    rng = np.random.default_rng(5)
    tasks = tasks[rng.permutation(len(tasks))]
    n8 = int(((tasks["h0"] != 300) & ((tasks["lqlen"] > 0) | (tasks["rqlen"] > 0))).sum())
    for variant, gaps in ((0, {}), (1, dict(o_del=5, e_del=2, o_ins=7, e_ins=1))):
        p = host.default_params(variant=variant, w=spec.get("w", 100), **gaps)
        order, seg, words = host.plan_batch(p, tasks, kernel=host.KERNEL_AUTO)
        lane_mode = n8 > 49152                                            # (the 16-bit seeds then sit in the 16-bit lane class, not in the wave classes)
        w16 = tasks["h0"] == 300
        n16l, n16r = (int((w16 & (tasks["lqlen"] > 0)).sum()), int((w16 & (tasks["rqlen"] > 0)).sum())) if lane_mode else (0, 0)
        assert seg[17] - seg[9] == n8 + n16l and seg[25] - seg[17] == n16r          # a place for every 8-bit lane seed on the left lists, none on a right list
        left = order[seg[9]:seg[17]]
        left = left[left != 0xffffffff]                                   # (the seeds with an N in a query are on the general kernel's N list: bsw_binparams.nsplit)
        nsplit = int(n8 - len(np.unique(left[~w16[left]])))
        assert 0 < nsplit < n8 // 2
        want = oracle.pair_batch(p, tasks, nthreads=8)
        with host.BswContext(device=0) as c:
            b = c.upload(p, tasks); c.run(b); got = c.download(b)
            nwave = int(sum(1 for k in range(8) if seg[k + 1] - seg[k]))
            assert b.info()["launches"] == nwave + 3 + (2 + 1 if lane_mode else 0)       # the general classes, the N list, ONE fused launch, the redo launch (+ the 16-bit class: two sides, bsw_pair_finalize)
            b.free()
            assert_same(got, want, tasks)
            assert_same(c.extend_pairs(p, tasks), want, tasks)            # the same through a submit (one chunk)
            pt, pwords = host.pack_tasks(tasks)
            assert_same(c.extend_pairs_packed(p, pt), want, tasks)
        with host.BswContext(device=0, result_format=host.RESULT_PAIR) as c:
            gp = c.extend_pairs(p, tasks)
            for f in ("tag", "qb", "qe", "rb", "re", "score", "truesc", "w"):
                assert (gp[f] == want[f]).all(), f


@pytest.mark.parametrize("n", [20_000, 60_000])
def test_queries_with_an_n_leave_the_lane_lists_of_a_chunk_that_does_not_fill_the_machine(host, oracle, n):
    """bsw_binparams.nsplit (BSW_KERNEL_AUTO, up to NSPLIT_MAX lane seeds, not for chunks of a streaming submit): a launch that does
    not fill the machine lasts as long as its slowest wavefront, and a wavefront of the two-seeds-per-lane kernels whose queries
    hold Ns runs 1.6 - 2x as long as the others.  The 8-bit lane seeds with an N in a query go on a list of their own for the
    general kernel (beside the lane launches); the lane lists keep their counted sizes, unused tails marked.  One-sided seeds
    (nothing to fuse): 20 000 run the group kernel, 60 000 the lane kernels.  Bytes, packed and pair-record paths."""
    tasks, arena = host.synth_tasks(n, seed=23, sub_rate=0.02, indel_rate=0.005, junk_frac=0.05, n_rate=0.003 if n < 50_000 else 0.001)     # (the list's work is capped: NLIST_WORK_MAX)
    p = host.default_params()
    order, seg, _ = host.plan_batch(p, tasks, kernel=host.KERNEL_AUTO)
    right = order[seg[17]:seg[25]]
    hn = _gen.query_has_n(tasks, arena, 1)
    kept = right[right != 0xffffffff]
    assert 0 < hn.sum() < n // 2 and len(kept) == n - int(hn.sum()) and not hn[kept].any()
    want = oracle.pair_batch(p, tasks, nthreads=8)
    with host.BswContext(device=0) as c:
        b = c.upload(p, tasks); c.run(b); got = c.download(b)
        assert b.info()["launches"] == 3                                  # the N list, the right sides, the redo launch
        b.free()
        assert_same(got, want, tasks)
        assert_same(c.extend_pairs(p, tasks), want, tasks)
        pt, pwords = host.pack_tasks(tasks)
        assert_same(c.extend_pairs_packed(p, pt), want, tasks)
    with host.BswContext(device=0, result_format=host.RESULT_PAIR) as c:
        gp = c.extend_pairs(p, tasks)
        for f in ("tag", "qb", "qe", "rb", "re", "score", "truesc", "w"):
            assert (gp[f] == want[f]).all(), f


def test_launch_chain_runs_beside_the_n_list(host, oracle):
    """131 072 seeds of 250 bp reads under BSW_KERNEL_AUTO: past the group kernel's fused range, so the lane kernels run their four
    chained launches (136- and 232-column classes, left and right) — and beside them the general kernel runs the seeds with an N
    in a query from their own list (bsw_binparams.nsplit), on the second borrowed stream.  Resident batch and one-chunk submit."""
    n = 131_072
    tasks, arena = host.synth_tasks(n, seed=61, read_len=250, seed_len_min=19, seed_len_max=40, seed_at_start=0,
                                    sub_rate=0.04, indel_rate=0.01, junk_frac=0.05, n_rate=0.0001, w=500)
    p = host.default_params(w=500)
    order, seg, _ = host.plan_batch(p, tasks, kernel=host.KERNEL_AUTO)
    lists = order[seg[9]:seg[25]]
    nn = int((lists == 0xffffffff).sum())
    assert int(seg[25] - seg[17]) > 0 and 0 < nn < n // 10              # a list per side (not fused), some seeds moved to the N list
    want = oracle.pair_batch_avx2(p, tasks, nthreads=8)
    with host.BswContext(device=0) as c:
        b = c.upload(p, tasks); c.run(b); got = c.download(b)
        assert b.info()["launches"] >= 6                                  # the N list, four lane launches, the redo launch
        for _ in range(3):
            c.run(b)
        again = c.download(b)
        b.free()
        assert_same(got, want, tasks)
        assert again.tobytes() == got.tobytes()
        assert c.chain_timeouts() == 0
        assert_same(c.extend_pairs(p, tasks), want, tasks)
