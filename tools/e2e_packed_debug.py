#!/usr/bin/env python3
"""bsw_submit vs bsw_submit_packed on the bench workload with the library's per-slot timing lines (BSW_DEBUG_TIMING=1)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
import bench
host = graft.load_package().host
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
spec = dict(bench.WORKLOADS["150bp_w100_single_bin"])
p = host.default_params(w=spec["w"])
ha = host.HostArena(host.synth_arena_bound(n, **spec) + 4096)
tasks, _ = host.synth_tasks(n, arena=ha.u8, seed=1000, **spec)
need = int(host.lib().bsw_pack_tasks_bound(tasks.ctypes.data, len(tasks)))
pa = host.HostArena(need + 64)
pt, _w = host.pack_tasks(tasks, pa.view(np.uint64, need // 8 + 1))
ho = host.HostArena(n * host.RESULT.itemsize)
out = ho.view(host.RESULT, n)
cfgs = [(4, 131072)] if len(sys.argv) < 3 else [tuple(int(x) for x in a.split(":")) for a in sys.argv[2:]]   # streams:chunk[:pack_threads]
for cfg in cfgs:
    streams, chunk = cfg[0], cfg[1]
    pth = cfg[2] if len(cfg) > 2 else 4
    with host.BswContext(device=0, streams=streams, pack_threads=pth, chunk_tasks=chunk) as c:
        for name, fn, t in (("bytes", c.extend_pairs, tasks), ("packed", c.extend_pairs_packed, pt)):
            fn(p, t, out=out)
            best = 1e9
            for rep in range(5):
                t0 = time.perf_counter(); fn(p, t, out=out); dt = time.perf_counter() - t0
                best = min(best, dt)
            print("streams %d chunk %d pack_threads %d %s: best of 5 %.2f ms = %.1f M seeds/s" % (streams, chunk, pth, name, best * 1e3, n / best / 1e6), flush=True)
