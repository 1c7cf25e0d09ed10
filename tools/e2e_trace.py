#!/usr/bin/env python3
"""One warm-up + N timed bsw_submit passes over a registered arena (for rocprofv3 --kernel-trace --memory-copy-trace).
Fourth argument "ref": bsw_submit_ref against a 64 Mbp device-resident synthetic genome instead; "packed": bsw_submit_packed."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
host = graft.load_package().host
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
streams = int(sys.argv[2]) if len(sys.argv) > 2 else 4
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
refmode = len(sys.argv) > 4 and sys.argv[4] == "ref"
p = host.default_params()
ha = host.HostArena(host.synth_arena_bound(n) + 4096)
ho = host.HostArena(n * host.RESULT.itemsize)
tasks, _ = host.synth_tasks(n, arena=ha.u8, seed=1000)
obuf = ho.view(host.RESULT, n)
if refmode:
    lp = 64_000_000
    pac, rt, _ = host.synth_ref_tasks(n, lp, p, arena=ha.u8, seed=3000, read_len=150, seed_len_min=19, seed_len_max=19, seed_at_start=1)
    with host.BswContext(device=0, streams=streams, chunk_tasks=chunk) as ctx:
        ref = ctx.ref_upload(pac, lp)
        for k in range(3):
            t0 = time.perf_counter()
            ctx.submit_ref(p, ref, rt, out=obuf); ctx.wait()
            print("pass %.3f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
            time.sleep(0.01)
        ctx.ref_free(ref)
    sys.exit(0)
packed = len(sys.argv) > 4 and sys.argv[4] == "packed"
if packed:
    need = int(host.lib().bsw_pack_tasks_bound(tasks.ctypes.data, len(tasks)))
    pa = host.HostArena(need + 64)
    tasks, _w = host.pack_tasks(tasks, pa.view(np.uint64, need // 8 + 1))
with host.BswContext(device=0, streams=streams, chunk_tasks=chunk) as ctx:
    fn = ctx.extend_pairs_packed if packed else ctx.extend_pairs
    fn(p, tasks, out=obuf)
    for _ in range(2):
        time.sleep(0.01)
        t0 = time.perf_counter()
        fn(p, tasks, out=obuf)
        print("pass %.3f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
