"""GPU parity: HIP path (through the C ABI) vs the CPU oracle, bit-exact on every field."""
import numpy as np
import pytest

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
