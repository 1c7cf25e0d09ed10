#!/usr/bin/env python3
"""bsw_run (DP kernels only) against bsw_run_staged (pack + bin + DP kernels) on resident batches of the bench workloads;
results of both must be equal.  python tools/staged_rate.py [--tasks N] [--workloads a,b]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tasks", type=int, default=1_000_000)
    ap.add_argument("--workloads", default="150bp_w100_single_bin,150bp_w100_mixed_bins,250bp_w500")
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    host = graft.load_package().host
    out = {}
    with host.BswContext(device=0) as ctx:
        for wl in args.workloads.split(","):
            spec = dict(bench.WORKLOADS[wl])
            p = host.default_params(w=spec["w"])
            ar = host.HostArena(host.synth_arena_bound(args.tasks, **spec) + 4096)
            t, _ = host.synth_tasks(args.tasks, arena=ar.u8, seed=2000, **spec)
            b = ctx.upload_raw(p, t)
            ctx.run(b); ctx.sync()
            r0 = ctx.download(b)
            ctx.run_staged(b); ctx.sync()
            r1 = ctx.download(b)
            ctx.run_history2()
            for _ in range(args.reps):
                ctx.run(b)
            for _ in range(args.reps):
                ctx.run_staged(b)
            h = ctx.run_history2()
            k = float(np.mean([x[0] for x in h[:args.reps]]))
            s = float(np.mean([x[0] for x in h[args.reps:]]))
            st = float(np.mean([x[1] for x in h[args.reps:]]))
            cells = bench.cells_of(r0)
            out[wl] = {"seeds": args.tasks, "kernels_ms": round(k, 3), "staged_ms": round(s, 3), "pack_bin_ms": round(st, 3),
                       "gcups_kernels": round(cells / k / 1e6, 1), "gcups_staged": round(cells / s / 1e6, 1),
                       "equal": bool(r0.tobytes() == r1.tobytes()), "launches": b.info()["launches"]}
            b.free()
            ar.free()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
