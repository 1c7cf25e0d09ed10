#!/usr/bin/env python3
"""BENCH_r05's pair-record leg ran 8.33 / 8.23 / 10.98 / 12.85 / 8.20 ms for the same 1 M-seed submit.
The same leg (150 bp single bin, 1 M seeds, packed input, BSW_RESULT_PAIR, 4 slots, 112 Ki chunks) 40 times with
BSW_DEBUG_TIMING=1: every rep's wall time, and the slot lines of the slow ones (stderr carries a marker per rep).
BSW_DEBUG_TIMING=1 python tools/diag/pair_outliers.py 2> slots.txt"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402
host = graft.load_package().host
n = 1_000_000
chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 114688
p = host.default_params()
ha = host.HostArena(host.synth_arena_bound(n) + 4096)
tasks, _ = host.synth_tasks(n, arena=ha.u8, seed=1000)
need = int(host.lib().bsw_pack_tasks_bound(tasks.ctypes.data, n))
pa = host.HostArena(need + 64)
ptasks, _ = host.pack_tasks(tasks, pa.view(np.uint64, need // 8 + 1))
for fmt, dt in ((host.RESULT_PAIR, host.PAIR), (host.RESULT_FULL, host.RESULT)):
    ho = host.HostArena(n * dt.itemsize)
    out = ho.view(dt, n)
    with host.BswContext(device=0, streams=4, pack_threads=4, chunk_tasks=chunk, result_format=fmt) as c:
        for _ in range(2):
            c.extend_pairs_packed(p, ptasks, out=out)
        runs = []
        for r in range(40):
            sys.stderr.write("=== %s rep %d\n" % (dt.names[-1], r)); sys.stderr.flush()
            t0 = time.perf_counter()
            c.extend_pairs_packed(p, ptasks, out=out)
            runs.append((time.perf_counter() - t0) * 1e3)
    ho.free()
    print(json.dumps({"record": "pair (32 B)" if fmt == host.RESULT_PAIR else "full (96 B)", "chunk": chunk, "ms": [round(x, 2) for x in runs],
                      "median": round(float(np.median(runs)), 2), "max_over_min": round(max(runs) / min(runs), 3),
                      "slow_reps": [i for i, x in enumerate(runs) if x > 1.15 * float(np.median(runs))]}))
