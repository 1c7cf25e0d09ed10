#!/usr/bin/env python3
"""In the latency-bound range (a lane launch lasts as long as its longest wave however few seeds it holds) — is the 72-column class
as its OWN launch beside the 136-column one worth it, run side by side on forked streams with the right sides waiting only for the
left launches that hold their seeds?  L72 (0.35 ms) || L136 (0.97), then R136 (long right sides: their left sides are short, so
they wait for L72 only) || R72.  PE mixed seeds, device-resident, lane kernels forced, median ms of bsw_run.
Run with: (default) | BSW_NARROW_SHARE=0 | BSW_FORK=1 BSW_NARROW_SHARE=0"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as graft
host = graft.load_package().host
p = host.default_params()
tasks, arena = host.synth_tasks(262144, seed=51, seed_len_min=19, seed_len_max=60, seed_at_start=0, junk_frac=0.05)
row = {"BSW_FORK": os.environ.get("BSW_FORK"), "BSW_NARROW_SHARE": os.environ.get("BSW_NARROW_SHARE")}
with host.BswContext(device=0, kernel=host.KERNEL_LANE) as ctx:
    for n in (26208, 52416, 104832, 262144):
        b = ctx.upload(p, tasks[:n])
        for _ in range(3):
            ctx.run(b)
        ctx.sync(); ctx.run_history()
        for _ in range(12):
            ctx.run(b)
        ctx.sync()
        row[str(n)] = round(float(np.median(ctx.run_history())), 4)
        b.free()
print(json.dumps(row))
