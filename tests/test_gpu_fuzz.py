"""Randomised parameter-space fuzz: both kernels (forced lane bins, forced wave-per-task) against the oracle over random
scoring, band, zdrop, clip, band-try and variant settings and random seed shapes.  Deterministic (seeded)."""
import numpy as np
import pytest

import _gen
from test_gpu_parity import assert_same

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("block", range(6))
def test_fuzz_parameters_and_shapes(host, oracle, block):
    rng = np.random.default_rng(9000 + block)
    with host.BswContext(device=0, kernel=host.KERNEL_LANE) as lctx, host.BswContext(device=0, kernel=host.KERNEL_WAVE) as wctx:
        for it in range(6):
            a = int(rng.integers(1, 6))
            b = int(rng.integers(0, 10))
            nsc = int(rng.integers(-5, a + 1))
            over = dict(o_del=int(rng.integers(0, 20)), e_del=int(rng.integers(1, 8)), o_ins=int(rng.integers(0, 20)),
                        e_ins=int(rng.integers(1, 8)), w=int(rng.choice([1, 3, 10, 30, 100, 300])),
                        zdrop=int(rng.choice([0, 1, 10, 50, 100, 1000])), pen_clip5=int(rng.integers(0, 15)),
                        pen_clip3=int(rng.integers(0, 15)), max_band_try=int(rng.integers(1, 4)), variant=int(rng.integers(0, 2)))
            p = host.default_params(**over)
            p["mat"][0] = host.bwa_matrix(a=a, b=b, n=nsc)
            qmax = int(rng.choice([20, 70, 134, 231, 300]))
            seeds = _gen.random_seeds(rng, 500, qmin=1, qmax=qmax, tfac=float(rng.choice([1.0, 1.5, 2.5])),
                                      sub=float(rng.choice([0.0, 0.02, 0.1])), indel=float(rng.choice([0.0, 0.01, 0.06])),
                                      junk=float(rng.choice([0.0, 0.2, 0.6])), nrate=float(rng.choice([0.0, 0.001, 0.05])),
                                      h0max=int(rng.choice([5, 60, 250, 2000])))
            tasks, arena = host.make_tasks(seeds)
            want = oracle.pair_batch(p, tasks, nthreads=8)
            assert_same(lctx.extend_pairs(p, tasks), want, tasks)
            assert_same(wctx.extend_pairs(p, tasks), want, tasks)


@pytest.mark.parametrize("block", range(8))
def test_fuzz_two_seeds_per_lane_kernel(host, oracle, block):
    """The packed kernel's own parameter space (both variants, shared or separate deletion / insertion penalties,
    0 >= N score >= -b): random scoring, band, z-drop, clip penalties and seed shapes on both sides of its 8-bit score
    bound, forced lane bins."""
    rng = np.random.default_rng(12000 + block)
    with host.BswContext(device=0, kernel=host.KERNEL_LANE) as lctx:
        for it in range(5):
            a = int(rng.integers(1, 5))
            b = int(rng.integers(0, 9))
            nsc = -int(rng.integers(0, b + 1))
            o, e = int(rng.integers(0, 16)), int(rng.integers(1, 7))
            kind = (block * 5 + it) % 4                      # H/sym, M/sym, H/asym, M/asym in turn: the four kernel instantiations
            oi, ei = (o, e) if kind < 2 else (int(rng.integers(0, 16)), int(rng.integers(1, 7)))
            over = dict(o_del=o, e_del=e, o_ins=oi, e_ins=ei, w=int(rng.choice([1, 2, 7, 20, 100, 300])),
                        zdrop=int(rng.choice([0, 1, 10, 50, 100, 1000])), pen_clip5=int(rng.integers(0, 15)),
                        pen_clip3=int(rng.integers(0, 15)), max_band_try=int(rng.integers(1, 4)), variant=kind & 1)
            p = host.default_params(**over)
            p["mat"][0] = host.bwa_matrix(a=a, b=b, n=nsc)
            seeds = _gen.random_seeds(rng, 1500, qmin=1, qmax=int(rng.choice([12, 60, 134])), tfac=float(rng.choice([1.0, 1.6, 2.4])),
                                      sub=float(rng.choice([0.0, 0.02, 0.08])), indel=float(rng.choice([0.0, 0.01, 0.05])),
                                      junk=float(rng.choice([0.0, 0.2])), nrate=float(rng.choice([0.0, 0.002, 0.04])), h0max=60)
            for s in seeds:                                  # scores up to, at and beyond the class bound h0 + qlen*a + b = 255
                tot = (len(s.get("lq", ())) + len(s.get("rq", ()))) * a
                s["h0"] = max(1, min(s["h0"] + int(rng.integers(0, 200)), 255 - b - tot + int(rng.integers(-3, 4))))
            tasks, arena = host.make_tasks(seeds)
            want = oracle.pair_batch(p, tasks, nthreads=8)
            assert_same(lctx.extend_pairs(p, tasks), want, tasks)
