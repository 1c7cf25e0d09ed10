#!/usr/bin/env python3
"""Host-buffers-in / host-buffers-out rate of bsw_submit (nibble packing + pinned staging + H2D + kernels + D2H),
i.e. the PCIe-inclusive figure DESIGN.md quotes next to the HBM-resident bench value."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
host = graft.load_package().host
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
p = host.default_params()
tasks, arena = host.synth_tasks(n, seed=1000)
out = []
for streams, chunk, threads in ((1, 65536, 8), (2, 65536, 8), (4, 65536, 8), (4, 65536, 16), (8, 32768, 16), (4, 131072, 16)):
    with host.BswContext(device=0, streams=streams, chunk_tasks=chunk, pack_threads=threads) as ctx:
        buf = np.ones(n, dtype=host.RESULT)         # result buffer owned and already touched by the host, as in a C caller
        ctx.extend_pairs(p, tasks, out=buf)         # warm up (staging allocations, code load)
        t0 = time.perf_counter()
        res = ctx.extend_pairs(p, tasks, out=buf)
        dt = time.perf_counter() - t0
    cells = int(res["left"]["cells"].astype(np.int64).sum() + res["right"]["cells"].astype(np.int64).sum())
    out.append(dict(streams=streams, chunk_tasks=chunk, pack_threads=threads, seconds=round(dt, 4),
                    seeds_per_s=round(n / dt), gcups=round(cells / dt / 1e9, 1)))
print(json.dumps(out))
