#!/usr/bin/env python3
"""Long randomised comparison of the extension path (bsw_extend_pairs through every kernel selection) with the oracle:
random scoring / gaps / band / z-drop / clip / band-try settings and both recurrence variants, three quarters of the rounds
inside the two-seeds-per-lane kernels' parameter space (shared and separate gap penalties; query lengths of both their
classes), plus the packed-input path.  Not part of the test suite (minutes); prints one line per round."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft
import _gen
host = graft.load_package().host
oracle = graft.load_oracle()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
F = ["tag", "qb", "qe", "rb", "re", "score", "truesc", "w"]
E = ["score", "qle", "tle", "gtle", "gscore", "max_off", "aw", "cells"]
tot = 0
ctxs = [host.BswContext(device=0, kernel=k) for k in (host.KERNEL_AUTO, host.KERNEL_LANE, host.KERNEL_WAVE)]
for r in range(rounds):
    rng = np.random.default_rng(7000 + r)
    a, b = int(rng.integers(1, 5)), int(rng.integers(0, 9))
    sym = r % 2 == 0
    o, e = int(rng.integers(0, 14)), int(rng.integers(1, 6))
    over = dict(o_del=o, e_del=e, o_ins=o if sym else int(rng.integers(0, 14)), e_ins=e if sym else int(rng.integers(1, 6)),
                w=int(rng.choice([1, 3, 10, 33, 100, 400])), zdrop=int(rng.choice([0, 5, 50, 100, 1000])), pen_clip5=int(rng.integers(0, 12)),
                pen_clip3=int(rng.integers(0, 12)), max_band_try=int(rng.integers(1, 4)), variant=int(rng.integers(0, 2)))
    p = host.default_params(**over)
    p["mat"][0] = host.bwa_matrix(a=a, b=b, n=-int(rng.integers(0, b + 1)) if r % 4 != 3 else int(rng.integers(-5, a + 1)))
    seeds = _gen.random_seeds(rng, 20000, qmin=1, qmax=int(rng.choice([40, 134, 134, 231, 400])), tfac=float(rng.choice([1.0, 1.7, 2.4])),
                              sub=float(rng.choice([0.0, 0.02, 0.08])), indel=float(rng.choice([0.0, 0.01, 0.05])), junk=float(rng.choice([0.0, 0.2])),
                              nrate=float(rng.choice([0.0, 0.002, 0.03])), h0max=int(rng.choice([30, 120, 250])))
    tasks, arena = host.make_tasks(seeds)
    want = oracle.pair_batch(p, tasks, nthreads=16)
    ptasks, pwords = host.pack_tasks(tasks)
    for ci, c in enumerate(ctxs + ctxs[:1]):
        got = c.extend_pairs(p, tasks) if ci < len(ctxs) else c.extend_pairs_packed(p, ptasks)
        for f in F:
            if not (got[f] == want[f]).all():
                print("MISMATCH round", r, f, over, flush=True); sys.exit(1)
        for side in ("left", "right"):
            for f in E:
                if not (got[side][f] == want[side][f]).all():
                    print("MISMATCH round", r, side, f, over, flush=True); sys.exit(1)
    tot += len(tasks)
    print("round", r, "ok:", tot, "seeds x 3 kernel selections + packed input;", "a", a, "b", b, over, flush=True)
print("extend fuzz ok:", tot, "seeds, every field and cell count identical in AUTO / forced-lane / forced-wave selection and through bsw_submit_packed")
