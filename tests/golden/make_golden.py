#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz with the CPU oracle (run from the repo root: python tests/golden/make_golden.py).

The reference ships no vectors, so these fixtures pin the ORACLE's behaviour at
the commit that produced them: inputs (params + seeds) and expected result batches.  They let
(a) the oracle be checked for regressions on CPU and (b) the HIP path be checked on the GPU box
without rebuilding anything.  Deterministic: fixed seeds, no timestamps.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft  # noqa: E402
import _gen  # noqa: E402


def pack_seeds(seeds):
    """Flatten seed dicts into arrays that np.savez can hold."""
    keys = ("lq", "lt", "rq", "rt")
    lens = np.array([[len(s.get(k, ())) for k in keys] for s in seeds], dtype=np.int32)
    flat = np.concatenate([np.asarray(s.get(k, ()), dtype=np.uint8) for s in seeds for k in keys] + [np.zeros(0, np.uint8)])
    meta = np.array([[s["h0"], s.get("init_score", -1), s.get("tag", i)] for i, s in enumerate(seeds)], dtype=np.int64)
    return lens, flat, meta


def main():
    pkg = graft.load_package()
    host, orc = pkg.host, graft.load_oracle()
    out = os.path.dirname(os.path.abspath(__file__))
    sets = {
        "mixed_small": dict(rng=1, n=160, kw=dict(qmax=140, indel=0.02, junk=0.2, nrate=0.004)),
        "long_query": dict(rng=2, n=24, kw=dict(qmin=200, qmax=700, tfac=1.6, indel=0.01, junk=0.1)),
        "indel_heavy": dict(rng=3, n=96, kw=dict(qmin=60, qmax=140, indel=0.08, sub=0.02, junk=0.0)),
        "tiny": dict(rng=4, n=96, kw=dict(qmin=1, qmax=8, tfac=3.0, junk=0.5, h0max=12)),
    }
    pars = {"H_z100": dict(variant=0, zdrop=100), "M_z100": dict(variant=1, zdrop=100),
            "H_z0_w10": dict(variant=0, zdrop=0, w=10), "M_asym": dict(variant=1, zdrop=30, o_del=5, e_del=2, o_ins=7, e_ins=1)}
    for name, sp in sets.items():
        seeds = _gen.random_seeds(np.random.default_rng(sp["rng"]), sp["n"], **sp["kw"])
        tasks, arena = host.make_tasks(seeds)
        lens, flat, meta = pack_seeds(seeds)
        blob = dict(lens=lens, flat=flat, meta=meta)
        for pname, over in pars.items():
            p = host.default_params(**over)
            blob["params_" + pname] = p.view(np.uint8)
            blob["expect_" + pname] = orc.pair_batch(p, tasks).view(np.uint8)
        np.savez_compressed(os.path.join(out, name + ".npz"), **blob)
        print(name, len(seeds), "seeds")


if __name__ == "__main__":
    main()
