#!/usr/bin/env python3
"""250 bp / w=500 resident-batch rate for two task seeds and two warm-up depths (why bench.py's other_workloads figure and
tools/bins_sweep.sh's differ)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
import bench
host = graft.load_package().host
n = 1_000_000
sp = dict(bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "250bp_w500"])
p = host.default_params(zdrop=100, w=sp["w"])
with host.BswContext(device=0) as ctx:
    for seed in (1000, 2000, 3000):
        t, a = host.synth_tasks(n, seed=seed, **sp)
        b = ctx.upload(p, t)
        for warm, reps in ((1, 3), (3, 10)):
            for _ in range(warm):
                ctx.run(b)
            ctx.sync(); ctx.run_history()
            for _ in range(reps):
                ctx.run(b)
            ctx.sync()
            h = ctx.run_history()
            r = ctx.download(b)
            cells = int(r["left"]["cells"].astype(np.int64).sum() + r["right"]["cells"].astype(np.int64).sum())
            print("seed %d warm %d reps %d: mean %.3f ms min %.3f max %.3f -> %.1f GCUPS (mean)" % (seed, warm, reps, np.mean(h), np.min(h), np.max(h), cells / np.mean(h) / 1e6), flush=True)
        b.free()
