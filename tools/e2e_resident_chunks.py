#!/usr/bin/env python3
"""What the GPU alone does with a submit's chunks: 1 M seeds uploaded as HBM-resident batches of `chunk` seeds, one context
(= one stream) per in-flight chunk, all launched at once — no PCIe, no host pass.  Against one resident batch of 1 M."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
host = graft.load_package().host
n = 1_000_000
p = host.default_params()
tasks, arena = host.synth_tasks(n, seed=1000)
def timed(nctx, chunk):
    ctxs = [host.BswContext(device=0, streams=1) for _ in range(nctx)]
    bs = []
    for k, lo in enumerate(range(0, n, chunk)):
        c = ctxs[k % nctx]
        bs.append((c, c.upload(p, tasks[lo:lo + chunk])))
    best = 1e9
    for _ in range(6):
        t0 = time.perf_counter()
        for c, b in bs: c.run(b)
        for c in ctxs: c.sync()
        best = min(best, time.perf_counter() - t0)
    for c, b in bs: b.free()
    for c in ctxs: c.close()
    return best
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"))
for nctx, chunk in [(1, n), (1, 98304), (2, 98304), (3, 98304), (4, 98304), (6, 98304), (8, 98304), (4, 262144), (4, 65536), (8, 65536)]:
    t = timed(nctx, chunk)
    print("streams %d chunk %7d: %.2f ms = %.1f M seeds/s" % (nctx, chunk, t * 1e3, n / t / 1e6), flush=True)
