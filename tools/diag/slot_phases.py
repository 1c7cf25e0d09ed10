#!/usr/bin/env python3
"""Where a slot thread's time goes during one big submit: BSW_DEBUG_TIMING lines of ONE steady-state submit (10 M PE seeds),
summed per phase.  BSW_DEBUG_TIMING=1 python tools/diag/slot_phases.py [bytes|packed] [slots] [pack_threads] 2> lines.txt"""
import os, re, sys, time, subprocess, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if os.environ.get("SLOT_CHILD"):
    import __graft_entry__ as graft
    import bench
    host = graft.load_package().host
    fmt, slots, pth = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    n = 10_000_000
    spec = dict(bench.WORKLOADS["150bp_w100_mixed_bins"])
    p = host.default_params()
    chunk = 131072
    ids = list(range((n + chunk - 1) // chunk))
    tasks, ha = bench.generate_seeds(host, spec, ids, [min(chunk, n - c * chunk) for c in ids], lambda c: 7000 + c, 32)
    ho = host.HostArena(n * host.RESULT.itemsize)
    out = ho.view(host.RESULT, n)
    if fmt == "packed":
        need = int(host.lib().bsw_pack_tasks_bound(tasks.ctypes.data, n))
        pa = host.HostArena(need + 64)
        tasks, _ = host.pack_tasks(tasks, pa.view(np.uint64, need // 8 + 1))
    with host.BswContext(device=0, streams=slots, pack_threads=pth) as c:
        sub = c.submit if fmt == "bytes" else c.submit_packed
        for _ in range(2):
            sub(p, tasks, out); c.wait()
        sys.stderr.write("=== timed\n"); sys.stderr.flush()
        t0 = time.perf_counter()
        sub(p, tasks, out); c.wait()
        dt = time.perf_counter() - t0
        sys.stderr.write("=== end %.3f ms\n" % (dt * 1e3)); sys.stderr.flush()
    print(json.dumps({"format": fmt, "slots": slots, "pack_threads": pth, "ms": round(dt * 1e3, 2), "Mseeds_s": round(n / dt / 1e6, 1)}))
else:
    for args in (sys.argv[1:4],) if len(sys.argv) > 3 else (["packed", "4", "4"], ["packed", "4", "12"], ["bytes", "4", "4"], ["bytes", "4", "12"]):
        r = subprocess.run([sys.executable, os.path.abspath(__file__)] + list(args), env=dict(os.environ, SLOT_CHILD="1", BSW_DEBUG_TIMING="1"), capture_output=True, text=True)
        lines = r.stderr.split("=== timed\n")[-1].splitlines()
        tot = {"wait staging": 0.0, "host pass": 0.0, "wait prev results": 0.0, "DMA turn + enqueue": 0.0}
        nch = 0
        for l in lines:
            m = re.search(r"wait staging ([\d.]+) ms, host pass ([\d.]+), wait prev results ([\d.]+), DMA turn \+ enqueue ([\d.]+)", l)
            if m:
                nch += 1
                for k, v in zip(tot, m.groups()):
                    tot[k] += float(v)
        print(r.stdout.strip(), "| chunks", nch, "| per-slot-thread sums (ms):", {k: round(v / int(args[1]), 1) for k, v in tot.items()})
