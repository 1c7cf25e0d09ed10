#!/usr/bin/env python3
"""PCIe-inclusive rate of the submit paths on the metric's workload (150 bp PE mixed bins) for several (slots, chunk) settings:
single submits and a stream kept two deep in ONE context (tickets).  python tools/e2e_pe_sweep.py [--tasks N] [--cfgs 4,131072;4,262144] [--formats bytes,packed]"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--tasks", type=int, default=4_000_000)
ap.add_argument("--workload", default="150bp_w100_mixed_bins")
ap.add_argument("--cfgs", default="4,131072;4,196608;4,262144;4,393216;3,262144;6,131072;8,131072")
ap.add_argument("--formats", default="bytes,packed")
ap.add_argument("--reps", type=int, default=4)
ap.add_argument("--pairs", action="store_true")
args = ap.parse_args()
host = graft.load_package().host
spec = dict(bench.WORKLOADS[args.workload])
p = host.default_params(w=spec["w"])
n = args.tasks
chunk = 131072
ids = list(range((n + chunk - 1) // chunk))
tasks, ha = bench.generate_seeds(host, spec, ids, [min(chunk, n - c * chunk) for c in ids], lambda c: 7000 + c, 32)
fmt = host.RESULT_PAIR if args.pairs else host.RESULT_FULL
odt = host.PAIR if args.pairs else host.RESULT
ho = host.HostArena(2 * n * odt.itemsize)
o1, o2 = ho.view(odt, n), ho.view(odt, n, offset=n * odt.itemsize)
need = int(host.lib().bsw_pack_tasks_bound(tasks.ctypes.data, len(tasks)))
pa = host.HostArena(need + 64)
ptasks, _ = host.pack_tasks(tasks, pa.view(np.uint64, need // 8 + 1))
for f in args.formats.split(","):
    for cfg in args.cfgs.split(";"):
        slots, ch = (int(x) for x in cfg.split(","))
        with host.BswContext(device=0, streams=slots, chunk_tasks=ch, result_format=fmt) as c:
            sub = (lambda o: c.submit(p, tasks, o)) if f == "bytes" else (lambda o: c.submit_packed(p, ptasks, o))
            for _ in range(2):
                sub(o1); c.wait()
            runs = []
            for _ in range(args.reps):
                t0 = time.perf_counter(); sub(o1); c.wait(); runs.append(time.perf_counter() - t0)
            s0 = c.host_stats()
            t0 = time.perf_counter()
            sub(o1); ta = c.last_ticket
            sub(o2); tb = c.last_ticket
            R = 4
            for _ in range(R - 1):
                c.wait_ticket(ta); sub(o1); ta = c.last_ticket
                c.wait_ticket(tb); sub(o2); tb = c.last_ticket
            c.wait_ticket(ta); c.wait_ticket(tb)
            d = (time.perf_counter() - t0) / (2 * R)
            s1 = c.host_stats()
            cpu = (s1["slot_cpu_ns"] + s1["helper_cpu_ns"] - s0["slot_cpu_ns"] - s0["helper_cpu_ns"]) / 1e9 / (2 * R * n) * 1e6
        print(json.dumps(dict(format=f, slots=slots, chunk=ch, single_Mseeds_s=round(n / float(np.median(runs)) / 1e6, 1),
                              single_ms=[round(x * 1e3, 1) for x in runs], stream_Mseeds_s=round(n / d / 1e6, 1), cpu_s_per_Mseed=round(cpu, 4))), flush=True)
