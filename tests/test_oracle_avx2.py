"""The strong CPU baseline (oracle/ksw_extend_avx2.c: inter-task AVX2, 16 seeds per register) must return the same
bytes as the scalar oracle — every field of every seed, cell counts included — or its GCUPS figure means nothing."""
import numpy as np
import pytest

import _gen


def same(oracle, p, tasks, nthreads=4):
    want = oracle.pair_batch(p, tasks, nthreads=nthreads)
    got = oracle.pair_batch_avx2(p, tasks, nthreads=nthreads)
    if got.tobytes() != want.tobytes():
        for f in want.dtype.names:
            if want.dtype[f].names:
                for g in want.dtype[f].names:
                    bad = np.nonzero(got[f][g] != want[f][g])[0]
                    assert bad.size == 0, "%s.%s task %s got %s want %s" % (f, g, bad[:4], got[f][g][bad[:4]], want[f][g][bad[:4]])
            else:
                bad = np.nonzero(got[f] != want[f])[0]
                assert bad.size == 0, "%s task %s got %s want %s" % (f, bad[:4], got[f][bad[:4]], want[f][bad[:4]])
        assert False, "bytes differ outside the named fields"


@pytest.mark.parametrize("over", [
    dict(), dict(zdrop=0), dict(variant=1), dict(w=5), dict(w=1, zdrop=0), dict(w=40, zdrop=30, variant=1),
    dict(o_del=4, e_del=2, o_ins=7, e_ins=1), dict(o_del=0, e_del=1, o_ins=0, e_ins=1, variant=1), dict(max_band_try=1),
    dict(max_band_try=3, w=8), dict(pen_clip5=0, pen_clip3=0), dict(pen_clip5=20, pen_clip3=1, variant=1),
])
def test_random_seeds(host, oracle, over):
    rng = np.random.default_rng(abs(hash(str(sorted(over.items())))) % (2 ** 31))
    seeds = _gen.random_seeds(rng, 1500, qmin=1, qmax=160, tfac=2.2, sub=0.03, indel=0.02, junk=0.15, nrate=0.004, h0max=200)
    tasks, arena = host.make_tasks(seeds)
    same(oracle, host.default_params(**over), tasks)


def test_bench_workloads(host, oracle):
    for spec in (dict(), dict(seed_len_min=19, seed_len_max=60, seed_at_start=0, junk_frac=0.05, n_rate=0.0005),
                 dict(read_len=250, seed_len_min=19, seed_len_max=40, seed_at_start=0, sub_rate=0.04, indel_rate=0.01, junk_frac=0.05,
                      n_rate=0.0005, w=500)):
        tasks, arena = host.synth_tasks(3000, seed=11, **spec)
        same(oracle, host.default_params(w=spec.get("w", 100)), tasks)


def test_host_band_limits_and_retries(host, oracle):
    tasks, arena = host.synth_tasks(2000, seed=12, seed_len_min=19, seed_len_max=60, seed_at_start=0, indel_rate=0.04)
    rng = np.random.default_rng(2)
    tasks["wlim_l"] = rng.choice([0, 1, 3, 9, 200], len(tasks))
    tasks["wlim_r"] = rng.choice([0, 2, 5, 12, 200], len(tasks))
    same(oracle, host.default_params(w=6, max_band_try=3), tasks)


def test_what_the_vector_lanes_cannot_hold_goes_scalar(host, oracle):
    """general matrices, long queries / wide scores, ragged and tiny batches"""
    rng = np.random.default_rng(3)
    seeds = _gen.random_seeds(rng, 300, qmin=1, qmax=600, tfac=1.5, sub=0.05, indel=0.02, junk=0.1, nrate=0.01, h0max=3000)
    tasks, arena = host.make_tasks(seeds)
    p = host.default_params()
    same(oracle, p, tasks)
    p["mat"][0] = rng.integers(-6, 6, 25).astype(np.int8)
    p["mat"][0][0] = 3
    same(oracle, p, tasks)
    for n in (1, 2, 15, 16, 17, 33):
        same(oracle, host.default_params(), tasks[:n], nthreads=1)
