#!/usr/bin/env python3
"""How the AVX2 CPU baseline scales with threads on this box, and what CPU time the box grants the process (cgroup quota, load):
python tools/diag/cpu_scaling.py [--seeds N]"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402
import bench  # noqa: E402


def rd(p):
    try:
        return open(p).read().strip()
    except OSError:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=2_000_000)
    a = ap.parse_args()
    host = graft.load_package().host
    orc = graft.load_oracle()
    spec = dict(bench.WORKLOADS["150bp_w100_mixed_bins"])
    p = host.default_params(w=spec["w"])
    t, arena = host.synth_tasks(a.seeds, seed=9, **spec)          # (the tasks point into the arena: keep it)
    out = {"affinity": len(os.sched_getaffinity(0)), "cpu_count": os.cpu_count(), "loadavg": rd("/proc/loadavg"),
           "cgroup_cpu_max": rd("/sys/fs/cgroup/cpu.max"), "cfs_quota_us": rd("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"),
           "cfs_period_us": rd("/sys/fs/cgroup/cpu/cpu.cfs_period_us"), "cpu_stat": rd("/sys/fs/cgroup/cpu.stat"), "rows": []}
    cells = None
    for nth in (1, 4, 8, 16, 32, 64, 128):
        if nth > out["affinity"]:
            break
        n = a.seeds if nth >= 8 else a.seeds // 8
        best = 1e9
        for _ in range(3):
            c0 = time.process_time(); t0 = time.perf_counter()
            r = orc.pair_batch_avx2(p, t[:n], nthreads=nth)
            dt = time.perf_counter() - t0; cpu = time.process_time() - c0
            best = min(best, dt)
        out["rows"].append({"threads": nth, "seeds": n, "s": round(best, 3), "gcups": round(bench.cells_of(r) / best / 1e9, 2), "cpu_s_last": round(cpu, 2), "wall_s_last": round(dt, 3)})
        print(out["rows"][-1], flush=True)
    out["cpu_stat_after"] = rd("/sys/fs/cgroup/cpu.stat")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
