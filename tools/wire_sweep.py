#!/usr/bin/env python3
"""Wire-format rate (bsw_refbatch_submit x 128 + bsw_refbatch_wait) by group size (BSW_REFBATCH_GROUP, read once per process:
run one process per value) and host threads.  Usage: wire_sweep.py <pack_threads> [batches] [registered]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
host = graft.load_package().host
p = host.default_params(zdrop=0)
pt = int(sys.argv[1]) if len(sys.argv) > 1 else 4
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 128
tasks, arena = host.synth_tasks(nb * 819, seed=51, seed_len_min=19, seed_len_max=60, seed_at_start=0, junk_frac=0.05)
reg = len(sys.argv) > 3 and sys.argv[3] == "registered"          # the task batches back to back in registered host memory (DMA'd where they are)
warena = host.HostArena(nb * host.REFBATCH_IN_WORDS * 4) if reg else None
wview = warena.view(np.uint32, nb * host.REFBATCH_IN_WORDS).reshape(nb, host.REFBATCH_IN_WORDS) if reg else None
oarena = host.HostArena(nb * host.REFBATCH_OUT_WORDS * 4) if reg else None      # registered: the result batches too
oview = oarena.view(np.uint32, nb * host.REFBATCH_OUT_WORDS).reshape(nb, host.REFBATCH_OUT_WORDS) if reg else None
ins, outs, lo = [], [], 0
while lo < len(tasks) and len(ins) < nb:
    w, n = host.refbatch_encode(p, tasks[lo:lo + 819])
    if reg:
        wview[len(ins)] = w; w = wview[len(ins)]
    ins.append(w); outs.append(oview[len(outs)] if reg else np.zeros(host.REFBATCH_OUT_WORDS, np.uint32)); lo += n
kern = int(os.environ.get("WIRE_KERNEL", "0"))                    # 2: lane bins forced whatever the group size
with host.BswContext(device=0, pack_threads=pt, kernel=kern) as c:
    ts = []
    for rep in range(7):
        t0 = time.perf_counter()
        for a, b in zip(ins, outs):
            c.refbatch_submit(a, b)
        c.refbatch_wait(0, 0)
        ts.append(time.perf_counter() - t0)
ts = sorted(ts[1:])
print(json.dumps({"kernel": kern, "input": "registered" if reg else "pageable", "group": os.environ.get("BSW_REFBATCH_GROUP", "default"), "pack_threads": pt, "batches": nb, "seeds": lo,
                  "ms_min_median": [round(ts[0] * 1e3, 3), round(ts[len(ts) // 2] * 1e3, 3)], "M_seeds_per_s_best": round(lo / ts[0] / 1e6, 2)}))
