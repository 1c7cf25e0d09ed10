#!/usr/bin/env python3
"""Wave-per-task vs lane-per-task kernel by batch size (device-resident, 150 bp single bin): where BSW_KERNEL_AUTO should switch."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
host = graft.load_package().host
p = host.default_params()
tasks, arena = host.synth_tasks(131072, seed=5)
out = []
for n in (256, 1024, 2048, 4096, 8192, 12288, 16384, 20480, 24576, 32768, 65536, 131072):
    row = {"seeds": n}
    for name, kern in (("wave_ms", host.KERNEL_WAVE), ("lane_ms", host.KERNEL_LANE)):
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
