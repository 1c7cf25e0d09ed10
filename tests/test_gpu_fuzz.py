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


QUAD_SNIPPET = r"""
import sys, numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import __graft_entry__ as g
import _gen
host, orc = g.load_package().host, g.load_oracle()
from test_gpu_parity import assert_same
rng = np.random.default_rng(4242)
with host.BswContext(device=0, kernel=host.KERNEL_WAVE) as c:
    for it in range(10):
        a, b = int(rng.integers(1, 6)), int(rng.integers(0, 10))
        over = dict(o_del=int(rng.integers(0, 20)), e_del=int(rng.integers(1, 8)), o_ins=int(rng.integers(0, 20)), e_ins=int(rng.integers(1, 8)),
                    w=int(rng.choice([1, 3, 10, 30, 100, 300])), zdrop=int(rng.choice([0, 1, 10, 50, 100, 1000])), pen_clip5=int(rng.integers(0, 15)),
                    pen_clip3=int(rng.integers(0, 15)), max_band_try=int(rng.integers(1, 4)), variant=it & 1)
        p = host.default_params(**over)
        if it %% 3:
            p["mat"][0] = host.bwa_matrix(a=a, b=b, n=int(rng.integers(-5, a + 1)))
        else:
            p["mat"][0] = rng.integers(-9, 10, 25).astype(np.int8)          # a general 5x5 matrix
        qmax = int(rng.choice([20, 63, 64, 127, 128, 191, 192, 255]))
        seeds = _gen.random_seeds(rng, 700, qmin=1, qmax=qmax, tfac=float(rng.choice([1.0, 1.5, 2.5])), sub=float(rng.choice([0.0, 0.02, 0.1])),
                                  indel=float(rng.choice([0.0, 0.01, 0.06])), junk=float(rng.choice([0.0, 0.2, 0.6])),
                                  nrate=float(rng.choice([0.0, 0.001, 0.05])), h0max=int(rng.choice([5, 60, 250, 2000])))
        tasks, arena = host.make_tasks(seeds)
        assert_same(c.extend_pairs(p, tasks), orc.pair_batch(p, tasks, nthreads=8), tasks)
    # targets longer than one 256-base refill, one to three seeds in the last wavefront
    for n in (1, 2, 3, 5, 64, 257):
        seeds = _gen.random_seeds(rng, n, qmin=100, qmax=250, tfac=3.0, sub=0.02, indel=0.01, junk=0.0, nrate=0.0, h0max=300)
        tasks, arena = host.make_tasks(seeds)
        p = host.default_params(w=300, zdrop=0)
        assert_same(c.extend_pairs(p, tasks), orc.pair_batch(p, tasks, nthreads=8), tasks)
# the redo list of the lane kernels (its length is counted on the device) through the same kernel
with host.BswContext(device=0, kernel=host.KERNEL_LANE) as c:
    for over in (dict(w=8, zdrop=0), dict(w=3), dict(variant=1, w=12, max_band_try=3)):
        p = host.default_params(**over)
        tasks, arena = host.synth_tasks(20000, seed=81, seed_len_min=19, seed_len_max=60, seed_at_start=0, indel_rate=0.02, junk_frac=0.1, n_rate=0.001)
        want = orc.pair_batch(p, tasks, nthreads=8)
        assert int((want["left"]["aw"] > p["w"][0]).sum() + (want["right"]["aw"] > p["w"][0]).sum()) > 20       # band retries happen
        assert_same(c.extend_pairs(p, tasks), want, tasks)
print("ok")
"""


def test_four_seeds_per_wavefront_kernel_forced():
    """bsw_quad_kernel on every general class up to 256 columns whatever the batch size (BSW_QUAD=1; by default only the
    192 / 256-column classes of batches >= 8 192 seeds take it): random scoring incl. general 5x5 matrices, both variants,
    band retries in-kernel, class boundaries, targets beyond one refill, ragged last wavefronts.  The switch is read once
    per process: own process."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", QUAD_SNIPPET % dict(root=root)], env=dict(os.environ, BSW_QUAD="1"),
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout[-2000:] + out.stderr[-4000:]


def test_four_seeds_per_wavefront_kernel_by_default(host, oracle):
    """>= 8 192 long seeds in one general-class launch take the four-seeds-per-wavefront kernel without any switch."""
    p = host.default_params()
    tasks, arena = host.synth_tasks(9000, seed=91)                        # 131 x 257: the 192-column class
    with host.BswContext(device=0, kernel=host.KERNEL_WAVE) as c:
        assert_same(c.extend_pairs(p, tasks), oracle.pair_batch(p, tasks, nthreads=8), tasks)
