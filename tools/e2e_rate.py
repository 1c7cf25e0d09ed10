#!/usr/bin/env python3
"""Host-buffers-in / host-buffers-out rate of bsw_submit for several slot / chunk settings: sequences and results in
registered (pinned) host memory (DMA direct, pack + bin on the GPU) and in pageable memory (host gather threads);
single submits (latency-inclusive) and a stream of submits kept two deep through two contexts (steady state).
The PCIe-inclusive figures DESIGN.md quotes next to the HBM-resident bench value."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
host = graft.load_package().host
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
p = host.default_params()
ha = host.HostArena(host.synth_arena_bound(n) + 4096)
ho = host.HostArena(2 * n * host.RESULT.itemsize)
tasks, _ = host.synth_tasks(n, arena=ha.u8, seed=1000)
obuf = ho.view(host.RESULT, n)
obuf2 = ho.view(host.RESULT, n, offset=n * host.RESULT.itemsize)
ptasks, parena = host.synth_tasks(n, seed=1000)
pout = np.ones(n, dtype=host.RESULT)

def cells_of(res):
    return int(res["left"]["cells"].astype(np.int64).sum() + res["right"]["cells"].astype(np.int64).sum())

CFGS = ((1, 4, 65536, 4), (1, 4, 131072, 4), (1, 8, 65536, 4), (1, 3, 131072, 4), (1, 4, 262144, 4), (1, 2, 262144, 4), (0, 4, 65536, 4), (0, 4, 131072, 4))
if len(sys.argv) > 2:
    CFGS = tuple(tuple(int(x) for x in a.split(",")) for a in sys.argv[2:])
for reg, streams, chunk, threads in CFGS:
    t, o = (tasks, obuf) if reg else (ptasks, pout)
    with host.BswContext(device=0, streams=streams, chunk_tasks=chunk, pack_threads=threads) as ctx:
        ctx.extend_pairs(p, t, out=o)               # warm up (staging allocations, code load)
        best = 1e9
        for _ in range(4):
            t0 = time.perf_counter()
            res = ctx.extend_pairs(p, t, out=o)
            best = min(best, time.perf_counter() - t0)
    print(json.dumps(dict(mode="single submit", registered=bool(reg), streams=streams, chunk_tasks=chunk, pack_threads=threads, seconds=round(best, 5),
                          seeds_per_s=round(n / best), gcups=round(cells_of(res) / best / 1e9, 1))), flush=True)

# steady state: a stream of 1M-seed batches, two in flight (two contexts, `streams` slot threads each)
for streams, chunk in (((2, 131072), (2, 65536), (3, 131072), (4, 131072), (2, 262144)) if len(sys.argv) <= 2 else ()):
    a = host.BswContext(device=0, streams=streams, chunk_tasks=chunk)
    b = host.BswContext(device=0, streams=streams, chunk_tasks=chunk)
    a.extend_pairs(p, tasks, out=obuf); b.extend_pairs(p, tasks, out=obuf2)
    reps = 8
    t0 = time.perf_counter()
    a.submit(p, tasks, obuf); b.submit(p, tasks, obuf2)
    for _ in range(reps - 1):
        a.wait(); a.submit(p, tasks, obuf)
        b.wait(); b.submit(p, tasks, obuf2)
    a.wait(); b.wait()
    dt = (time.perf_counter() - t0) / (2 * reps)
    same = bool(obuf.tobytes() == obuf2.tobytes())
    print(json.dumps(dict(mode="stream of submits, two in flight", streams_per_ctx=streams, slot_threads=2 * streams, chunk_tasks=chunk,
                          seconds_per_batch=round(dt, 5), seeds_per_s=round(n / dt), gcups=round(cells_of(obuf) / dt / 1e9, 1), identical=same)), flush=True)
    a.close(); b.close()
