#!/usr/bin/env python3
"""What the GPU alone does with the chunks of a PE mixed-bin submit: 2 M seeds uploaded as HBM-resident batches of `chunk`
seeds (staged: pack + bin + DP per run, as a submit's chunk runs them), one context (= one stream) per in-flight chunk, all
launched at once — no PCIe, no host pass.  Against one resident batch of 2 M."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
import bench
host = graft.load_package().host
n = 2_000_000
spec = dict(bench.WORKLOADS["150bp_w100_mixed_bins"])
p = host.default_params()
ar = host.HostArena(host.synth_arena_bound(n, **spec) + 4096)
tasks, _ = host.synth_tasks(n, arena=ar.u8, seed=1000, **spec)
def timed(nctx, chunk, staged=True):
    ctxs = [host.BswContext(device=0, streams=1) for _ in range(nctx)]
    bs = []
    for k, lo in enumerate(range(0, n, chunk)):
        c = ctxs[k % nctx]
        bs.append((c, c.upload_raw(p, tasks[lo:lo + chunk])))
    best = 1e9
    for _ in range(6):
        t0 = time.perf_counter()
        for c, b in bs: (c.run_staged if staged else c.run)(b)
        for c in ctxs: c.sync()
        best = min(best, time.perf_counter() - t0)
    for c, b in bs: b.free()
    for c in ctxs: c.close()
    return best
for nctx, chunk in [(1, n), (1, 131072), (2, 131072), (4, 131072), (8, 131072), (1, 262144), (2, 262144), (4, 262144), (2, 524288), (4, 524288)]:
    t = timed(nctx, chunk)
    print("streams %d chunk %7d: %.2f ms = %.1f M seeds/s" % (nctx, chunk, t * 1e3, n / t / 1e6), flush=True)
