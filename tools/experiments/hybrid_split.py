#!/usr/bin/env python3
"""What would routing the seeds with a LONG side to the general kernel, beside lane launches of the short ones, be worth in the
latency-bound range (a lane launch lasts as long as its longest wave, however few seeds)?  PE mixed seeds, device-resident: two
contexts (one forced to the lane kernels with the seeds whose longer side is <= T, one forced to the general kernel with the
rest) run from two threads at once, against either kernel alone on all seeds."""
import json, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as graft
host = graft.load_package().host
p = host.default_params()
tasks, arena = host.synth_tasks(262144, seed=51, seed_len_min=19, seed_len_max=60, seed_at_start=0, junk_frac=0.05)
qm_all = np.maximum(tasks["lqlen"], tasks["rqlen"])

def timed(fn, reps=12):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return round(float(np.median(ts)) * 1e3, 4)

rows = []
cl = host.BswContext(device=0, kernel=host.KERNEL_LANE)
cg = host.BswContext(device=0, kernel=host.KERNEL_WAVE)
for n in (16384, 32768, 52416, 104832, 262144):
    t = tasks[:n]; qm = qm_all[:n]
    row = {"seeds": n}
    b = cl.upload(p, t); row["lane_ms"] = timed(lambda: (cl.run(b), cl.sync())); b.free()
    b = cg.upload(p, t); row["general_ms"] = timed(lambda: (cg.run(b), cg.sync())); b.free()
    for T in (56, 71, 88, 104):
        short, long_ = t[qm <= T], t[qm > T]
        bs, bl = cl.upload(p, short), cg.upload(p, long_)
        def both():
            th = threading.Thread(target=lambda: (cg.run(bl), cg.sync()))
            th.start(); cl.run(bs); cl.sync(); th.join()
        row["T%d" % T] = {"long_share": round(len(long_) / n, 3), "both_ms": timed(both),
                          "lane_part_ms": timed(lambda: (cl.run(bs), cl.sync())), "general_part_ms": timed(lambda: (cg.run(bl), cg.sync()))}
        bs.free(); bl.free()
    rows.append(row)
    print(json.dumps(row), flush=True)
cl.close(); cg.close()
