#!/usr/bin/env python3
"""One warm-up + N timed bsw_submit passes over a registered arena (for rocprofv3 --kernel-trace --memory-copy-trace)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
host = graft.load_package().host
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
streams = int(sys.argv[2]) if len(sys.argv) > 2 else 4
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
p = host.default_params()
ha = host.HostArena(host.synth_arena_bound(n) + 4096)
ho = host.HostArena(n * host.RESULT.itemsize)
tasks, _ = host.synth_tasks(n, arena=ha.u8, seed=1000)
obuf = ho.view(host.RESULT, n)
with host.BswContext(device=0, streams=streams, chunk_tasks=chunk) as ctx:
    ctx.extend_pairs(p, tasks, out=obuf)
    for _ in range(2):
        t0 = time.perf_counter()
        ctx.extend_pairs(p, tasks, out=obuf)
        print("pass %.3f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
