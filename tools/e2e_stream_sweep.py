#!/usr/bin/env python3
"""A stream of packed submits kept two deep (two contexts): sweep of slots per context x chunk size."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
import bench
host = graft.load_package().host
n = 1_000_000
spec = dict(bench.WORKLOADS["150bp_w100_single_bin"])
p = host.default_params(w=spec["w"])
ha = host.HostArena(host.synth_arena_bound(n, **spec) + 4096)
tasks, _ = host.synth_tasks(n, arena=ha.u8, seed=1000, **spec)
need = int(host.lib().bsw_pack_tasks_bound(tasks.ctypes.data, len(tasks)))
pa = host.HostArena(need + 64)
pt, _w = host.pack_tasks(tasks, pa.view(np.uint64, need // 8 + 1))
ho = [host.HostArena(n * host.RESULT.itemsize) for _ in range(3)]
outs = [h.view(host.RESULT, n) for h in ho]
mode = sys.argv[1] if len(sys.argv) > 1 else "packed"
src = pt if mode == "packed" else tasks
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"), flush=True)
for nctx, streams, chunk in [(2, 2, 262144), (2, 2, 131072), (2, 3, 131072), (2, 2, 196608), (3, 2, 131072), (2, 4, 98304), (2, 4, 131072), (2, 3, 98304), (1, 8, 98304), (1, 6, 131072)]:
    ctxs = [host.BswContext(device=0, streams=streams, pack_threads=2, chunk_tasks=chunk) for _ in range(nctx)]
    sub = [(lambda c=c, o=o: (c.submit_packed(p, src, o) if mode == "packed" else c.submit(p, src, o))) for c, o in zip(ctxs, outs)]
    for c, s in zip(ctxs, sub):
        s(); c.wait()
    reps = 6
    t1 = time.perf_counter()
    for s in sub:
        s()
    for _ in range(reps - 1):
        for c, s in zip(ctxs, sub):
            c.wait(); s()
    for c in ctxs:
        c.wait()
    d = (time.perf_counter() - t1) / (nctx * reps)
    print("%s: %d contexts x %d slots, chunk %d: %.2f ms per batch = %.1f M seeds/s" % (mode, nctx, streams, chunk, d * 1e3, n / d / 1e6), flush=True)
    for c in ctxs:
        c.close()
