#!/usr/bin/env python3
"""A/B of two library builds (BSW_LIB_PATH) on resident PE batches with query Ns: median and min of 15 runs per size.
python tools/diag/ab_nfirst.py <mode: group|lane>"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
host = graft.load_package().host
p = host.default_params()
spec = dict(seed_len_min=19, seed_len_max=60, seed_at_start=0, junk_frac=0.05, n_rate=float(os.environ.get("N_RATE", "0.001")))
tasks, arena = host.synth_tasks(262144, seed=51, **spec)
row = {}
sizes = (24576, 32768, 49152) if sys.argv[1] == "group" else (65536, 131072, 262144)
for n in sizes:
    with host.BswContext(device=0, kernel=0) as ctx:
        b = ctx.upload(p, tasks[:n])
        for _ in range(3):
            ctx.run(b)
        ctx.sync(); ctx.run_history()
        for _ in range(15):
            ctx.run(b)
        ctx.sync()
        h = ctx.run_history()
        row[n] = [round(float(np.median(h)), 4), round(float(np.min(h)), 4)]
        b.free()
print(json.dumps(row))
