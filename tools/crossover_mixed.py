#!/usr/bin/env python3
"""crossover.py on the PE mixed-bin seeds (both sides, query lengths 0..131): general kernel (BSW_QUAD=0: one wavefront per
seed; default: four seeds per wavefront) vs lane kernels by batch size, device-resident."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
host = graft.load_package().host
p = host.default_params()
tasks, arena = host.synth_tasks(131072, seed=51, seed_len_min=19, seed_len_max=60, seed_at_start=0, junk_frac=0.05)
out = []
for n in (1024, 4096, 8192, 13104, 16384, 24576, 32768, 65536, 131072):
    row = {"seeds": n}
    for name, kern in (("general_ms", host.KERNEL_WAVE), ("lane_ms", host.KERNEL_LANE)):
        with host.BswContext(device=0, kernel=kern) as ctx:
            b = ctx.upload(p, tasks[:n])
            for _ in range(3):
                ctx.run(b)
            ctx.sync(); ctx.run_history()
            for _ in range(10):
                ctx.run(b)
            ctx.sync()
            row[name] = round(float(np.median(ctx.run_history())), 4)
            b.free()
    out.append(row)
print(json.dumps(out))
