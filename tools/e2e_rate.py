#!/usr/bin/env python3
"""Host-buffers-in / host-buffers-out rate of bsw_submit for several slot / chunk settings: sequences and results in
registered (pinned) host memory (DMA direct, pack + bin on the GPU) and in pageable memory (host gather threads).
The PCIe-inclusive figures DESIGN.md quotes next to the HBM-resident bench value."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
host = graft.load_package().host
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
p = host.default_params()
ha = host.HostArena(host.synth_arena_bound(n) + 4096)
ho = host.HostArena(n * host.RESULT.itemsize)
tasks, _ = host.synth_tasks(n, arena=ha.u8, seed=1000)
obuf = ho.view(host.RESULT, n)
ptasks, parena = host.synth_tasks(n, seed=1000)
pout = np.ones(n, dtype=host.RESULT)
out = []
for reg, streams, chunk, threads in ((1, 4, 65536, 4), (1, 4, 32768, 4), (1, 8, 32768, 4), (1, 3, 65536, 4), (1, 2, 131072, 4), (1, 4, 131072, 4), (1, 8, 65536, 4),
                                     (1, 4, 262144, 4), (1, 6, 49152, 4), (0, 4, 65536, 4), (0, 4, 131072, 4), (0, 8, 65536, 16)):
    t, o = (tasks, obuf) if reg else (ptasks, pout)
    with host.BswContext(device=0, streams=streams, chunk_tasks=chunk, pack_threads=threads) as ctx:
        ctx.extend_pairs(p, t, out=o)               # warm up (staging allocations, code load)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            res = ctx.extend_pairs(p, t, out=o)
            best = min(best, time.perf_counter() - t0)
    cells = int(res["left"]["cells"].astype(np.int64).sum() + res["right"]["cells"].astype(np.int64).sum())
    out.append(dict(registered=bool(reg), streams=streams, chunk_tasks=chunk, pack_threads=threads, seconds=round(best, 5),
                    seeds_per_s=round(n / best), gcups=round(cells / best / 1e9, 1)))
    print(json.dumps(out[-1]), flush=True)
