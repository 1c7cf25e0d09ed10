import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
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
streams = int(sys.argv[1]); chunk = int(sys.argv[2])
with host.BswContext(device=0, streams=streams, chunk_tasks=chunk) as ctx:
    for k in range(4):
        t0 = time.perf_counter()
        ctx.extend_pairs_packed(p, ptasks, out=obuf)
        print("pass %.2f ms" % ((time.perf_counter() - t0) * 1e3), file=sys.stderr, flush=True)
