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
pin = None if len(sys.argv) < 2 else (sys.argv[1] == "pin")
print("pin_threads =", pin, flush=True)
for streams, chunk in [(4, 98304), (3, 98304), (4, 98304), (3, 98304), (4, 131072), (3, 131072), (4, 98304), (3, 98304), (4, 131072), (3, 131072)]:
    with host.BswContext(device=0, streams=streams, chunk_tasks=chunk, pin_threads=pin) as ctx:
        if streams == 2:
            print("placement", ctx.placement(), "main thread may run on", len(os.sched_getaffinity(0)), "CPUs", flush=True)
        ctx.extend_pairs_packed(p, ptasks, out=obuf)
        ts = []
        for _ in range(7):
            t0 = time.perf_counter()
            ctx.extend_pairs_packed(p, ptasks, out=obuf)
            ts.append(time.perf_counter() - t0)
        ts.sort()
    print("slots %d chunk %6d: best %.2f ms median %.2f ms = %.1f M seeds/s (median)" % (streams, chunk, ts[0] * 1e3, ts[3] * 1e3, n / ts[3] / 1e6), flush=True)
