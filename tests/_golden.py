"""Loader for tests/golden/*.npz (see make_golden.py)."""
import glob
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def load(host, name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    lens, flat, meta = z["lens"], z["flat"], z["meta"]
    seeds, off = [], 0
    for i in range(len(lens)):
        s = {"h0": int(meta[i, 0]), "init_score": int(meta[i, 1]), "tag": int(meta[i, 2])}
        for k, key in enumerate(("lq", "lt", "rq", "rt")):
            n = int(lens[i, k])
            if key in ("lq", "rq") and n == 0:
                off += n
                continue
            if key in ("lt",) and "lq" not in s:
                off += n
                continue
            if key in ("rt",) and "rq" not in s:
                off += n
                continue
            s[key] = flat[off:off + n].copy()
            off += n
        seeds.append(s)
    tasks, arena = host.make_tasks(seeds)
    cases = {}
    for k in z.files:
        if k.startswith("params_"):
            pname = k[len("params_"):]
            cases[pname] = (z[k].view(host.PARAMS).copy(), z["expect_" + pname].view(host.RESULT).copy())
    return tasks, arena, cases
