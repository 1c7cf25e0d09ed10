#!/usr/bin/env python3
"""For rocprofv3 --kernel-trace: 1 M seeds as resident batches of 98304 on 4 contexts, launched at once, three times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
host = graft.load_package().host
n, chunk, nctx = 1_000_000, int(sys.argv[1]) if len(sys.argv) > 1 else 98304, int(sys.argv[2]) if len(sys.argv) > 2 else 4
p = host.default_params()
tasks, arena = host.synth_tasks(n, seed=1000)
ctxs = [host.BswContext(device=0, streams=1) for _ in range(nctx)]
bs = []
for k, lo in enumerate(range(0, n, chunk)):
    c = ctxs[k % nctx]
    bs.append((c, c.upload(p, tasks[lo:lo + chunk])))
for _ in range(3):
    t0 = time.perf_counter()
    for c, b in bs: c.run(b)
    for c in ctxs: c.sync()
    print("pass %.2f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
    time.sleep(0.02)
