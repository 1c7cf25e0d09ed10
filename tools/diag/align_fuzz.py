#!/usr/bin/env python3
"""Long randomised comparison of bsw_align_batch with the oracle (not part of the test suite: minutes of oracle time).
Small gap penalties and gap-rich pairs on purpose: they make the lazy-F loop run several rounds."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft
import _gen
host = graft.load_package().host
oracle = graft.load_oracle()
XB, XSTOP, XSUBO, XSTART = 0x10000, 0x20000, 0x40000, 0x80000
F = ("score", "te", "qe", "score2", "te2", "tb", "qb")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
tot = 0
with host.BswContext(device=0) as c:
    for r in range(rounds):
        rng = np.random.default_rng(5000 + r)
        a, b = int(rng.integers(1, 5)), int(rng.integers(0, 8))
        p = host.default_params(o_del=int(rng.integers(0, 8)), e_del=int(rng.integers(1, 4)), o_ins=int(rng.integers(0, 8)), e_ins=int(rng.integers(1, 4)))
        p["mat"][0] = host.bwa_matrix(a, b, -int(rng.integers(0, b + 1)))
        n = 4000
        keep, at = [], np.zeros(n, dtype=host.ATASK)
        for i in range(n):
            tl = int(rng.integers(1, 900)); t = rng.integers(0, 4, tl).astype(np.uint8)
            ql = int(rng.integers(1, 257))
            s0 = int(rng.integers(0, max(1, tl - ql)))
            q = _gen.mutate(rng, t[s0:s0 + ql], ql, float(rng.choice([0.0, 0.03, 0.1])), float(rng.choice([0.0, 0.02, 0.1]))) if rng.random() < 0.8 else rng.integers(0, 5, ql).astype(np.uint8)
            if len(q) == 0: q = np.zeros(1, np.uint8)
            q = np.ascontiguousarray(q[:256]); keep.append((q, t))
            at[i]["query"], at[i]["target"], at[i]["qlen"], at[i]["tlen"] = q.ctypes.data, t.ctypes.data, len(q), len(t)
            at[i]["xtra"] = int(rng.choice([XB, 0])) | int(rng.choice([0, XSTART, XSUBO | XSTART, XSUBO, XSTOP])) | int(rng.integers(0, 60))
        got = c.align_batch(p, at)
        want, _ = oracle.align2_batch(p["mat"][0], int(p["o_del"][0]), int(p["e_del"][0]), int(p["o_ins"][0]), int(p["e_ins"][0]), at, nthreads=16)
        for k, f in enumerate(F):
            bad = np.nonzero(got[f] != want[:, k])[0]
            if bad.size:
                print("MISMATCH round", r, f, bad[:5], got[f][bad[:5]], want[bad[:5], k], at["qlen"][bad[:5]], at["tlen"][bad[:5]], [hex(x) for x in at["xtra"][bad[:5]]], flush=True)
                sys.exit(1)
        tot += n
        print("round", r, "ok", tot, "alignments, a b", a, b, flush=True)
print("align fuzz ok:", tot, "alignments")
