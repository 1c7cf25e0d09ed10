#!/usr/bin/env python3
"""extend_fuzz.py at the EDGES of the two-seeds-per-lane kernels' parameter space (round 5: their row head / tail runs in packed
16-bit saturating arithmetic): gap penalties up to oe = 255, match scores up to 12, z-drop 0 / 1 / 65 535 / 70 000 / 2^30, bands 0 / 1 /
65 535 / 2^20, h0 up to the top of the 8-bit range, targets far longer than the band can follow, single-base queries.
Every field and the cell count against the oracle through AUTO / forced-lane / forced-general selection."""
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
    rng = np.random.default_rng(9000 + r)
    a = int(rng.choice([1, 1, 2, 5, 12]))
    b = int(rng.choice([0, 1, 4, 40, 120]))
    b = min(b, 255 - a)
    sym = r % 2 == 0
    e1, e2 = int(rng.choice([1, 2, 17, 120, 250])), int(rng.choice([1, 3, 60, 254]))
    o1, o2 = int(rng.integers(0, 256 - e1)), int(rng.integers(0, 256 - e2))
    over = dict(o_del=o1, e_del=e1, o_ins=o1 if sym else o2, e_ins=e1 if sym else e2,
                w=int(rng.choice([0, 1, 2, 100, 65535, 1 << 20])), zdrop=int(rng.choice([0, 1, 7, 100, 65535, 70000, 1 << 30])),
                pen_clip5=int(rng.choice([0, 5, 100, 100000])), pen_clip3=int(rng.choice([0, 5, 100, 100000])),
                max_band_try=int(rng.integers(1, 4)), variant=int(rng.integers(0, 2)))
    p = host.default_params(**over)
    p["mat"][0] = host.bwa_matrix(a=a, b=b, n=-int(rng.integers(0, b + 1)))
    qmax = int(rng.choice([1, 2, 20, 134, 231]))
    seeds = _gen.random_seeds(rng, 12000, qmin=1, qmax=qmax, tfac=float(rng.choice([0.3, 1.0, 2.4, 6.0])),
                              sub=float(rng.choice([0.0, 0.02, 0.3])), indel=float(rng.choice([0.0, 0.02, 0.2])), junk=float(rng.choice([0.0, 0.5])),
                              nrate=float(rng.choice([0.0, 0.01, 0.3])), h0max=int(rng.choice([1, 30, 120, 250])))
    tasks, arena = host.make_tasks(seeds)
    want = oracle.pair_batch(p, tasks, nthreads=16)
    for ci, c in enumerate(ctxs):
        got = c.extend_pairs(p, tasks)
        for f in F:
            if not (got[f] == want[f]).all():
                print("MISMATCH round", r, ci, f, over, a, b, flush=True); sys.exit(1)
        for side in ("left", "right"):
            for f in E:
                if not (got[side][f] == want[side][f]).all():
                    print("MISMATCH round", r, ci, side, f, over, a, b, flush=True); sys.exit(1)
    tot += len(tasks)
    print("round", r, "ok:", tot, "seeds x 3 kernel selections; a", a, "b", b, "qmax", qmax, over, flush=True)
print("extreme-parameter fuzz ok:", tot, "seeds, every field and cell count identical in AUTO / forced-lane / forced-general selection")
