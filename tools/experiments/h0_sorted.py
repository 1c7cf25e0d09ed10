#!/usr/bin/env python3
"""Would a SECOND sort key (the seed's h0) inside the query-length bins pay?  The lanes of a wave walk their rows in lockstep over
the union of their [beg, end) ranges; h0 sets how fast a seed's range opens and where its beg runs.  The device bins by query length only and
fills a bin in arrival order, which follows the task index closely — so handing the SAME seeds over sorted by h0 emulates the
second key.  PE mixed bins, 1 M seeds, device-resident, GCUPS from the exact cell count."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as graft
import bench
host = graft.load_package().host
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
spec = dict(bench.WORKLOADS[sys.argv[2] if len(sys.argv) > 2 else "150bp_w100_mixed_bins"])
w = spec.pop("w")
p = host.default_params(w=w)
tasks, arena = host.synth_tasks(n, seed=1000, w=w, **spec)
orders = {"as generated": np.arange(n), "sorted by h0": np.argsort(tasks["h0"], kind="stable"),
          "sorted by h0 descending": np.argsort(-tasks["h0"].astype(np.int64), kind="stable"),
          "sorted by h0 + lqlen": np.argsort(tasks["h0"].astype(np.int64) + tasks["lqlen"], kind="stable")}
with host.BswContext(device=0) as ctx:
    for name, o in orders.items():
        t = np.ascontiguousarray(tasks[o])
        b = ctx.upload(p, t)
        for _ in range(3):
            ctx.run(b)
        ctx.sync(); ctx.run_history()
        for _ in range(10):
            ctx.run(b)
        ctx.sync()
        ms = float(np.median(ctx.run_history()))
        r = ctx.download(b)
        cells = int(r["left"]["cells"].astype(np.int64).sum() + r["right"]["cells"].astype(np.int64).sum())
        print(json.dumps({"order": name, "ms": round(ms, 4), "gcups": round(cells / ms / 1e6, 1), "cells": cells}), flush=True)
        b.free()
