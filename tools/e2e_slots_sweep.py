#!/usr/bin/env python3
"""bsw_submit_packed of 1 M seeds (registered arena, results into registered memory): slots x chunk size, best of 5.
Run once plainly and once with GPU_MAX_HW_QUEUES=8 in the environment (the HIP runtime maps streams onto 4 hardware
queues by default: more than 4 slot streams then share queues and wait for each other's commands)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
host = graft.load_package().host
n = 1_000_000
p = host.default_params()
ha = host.HostArena(host.synth_arena_bound(n) + 4096)
ho = host.HostArena(n * host.RESULT.itemsize)
tasks, _ = host.synth_tasks(n, arena=ha.u8, seed=1000)
obuf = ho.view(host.RESULT, n)
need = int(host.lib().bsw_pack_tasks_bound(tasks.ctypes.data, len(tasks)))
pa = host.HostArena(need + 64)
ptasks, _w = host.pack_tasks(tasks, pa.view(np.uint64, need // 8 + 1))
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"), flush=True)
for streams, chunk in [(4, 98304), (5, 98304), (6, 98304), (8, 98304), (6, 65536), (8, 65536), (8, 49152), (6, 131072)]:
    with host.BswContext(device=0, streams=streams, chunk_tasks=chunk) as ctx:
        ctx.extend_pairs_packed(p, ptasks, out=obuf)
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            ctx.extend_pairs_packed(p, ptasks, out=obuf)
            best = min(best, time.perf_counter() - t0)
    print("slots %d chunk %6d: %.2f ms = %.1f M seeds/s" % (streams, chunk, best * 1e3, n / best / 1e6), flush=True)
