#!/usr/bin/env python3
"""profiles/pmc_latest.json: the counters bench.py quotes in its `roofline` objects, one entry per workload, from a
tools/profile.sh / tools/profile_quick.sh output directory (separate rocprofv3 --pmc passes).  HBM bytes per
MI355X_MICROARCH.md's gfx950 correction: FETCH_SIZE (KiB) x 2 + WRITE_SIZE (KiB), summed over the DP kernels of one bsw_run
step.  A step may launch the dominant kernel more than once (PE workloads: left sides, right sides): pmc_summary.py's values
are means per launch, so they are scaled by launches per step = calls / steps profiled.
Usage: make_pmc_latest.py <dir | summary.json> <workload> <seeds> <cells per step> [steps profiled = 4] [key in the file = workload]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = sys.argv[1]
workload, seeds, cells = sys.argv[2], int(sys.argv[3]), int(float(sys.argv[4]))
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 4           # bench.py --steps 3 --warmup 1
if d.endswith(".json"):                  # a kept tools/pmc_summary.py output instead of the raw rocprofv3 directories
    allk = json.load(open(d))
else:
    allk = json.loads(subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), d, "bsw"]))
step = {k: v for k, v in allk.items() if any(x in k for x in ("lane2_kernel", "lane2l_kernel", "lane2g_kernel", "lane_kernel", "pair_finalize", "wave_kernel", "quad_kernel", "long_kernel"))}
per_step = lambda v: max(1.0, v.get("calls", steps) / float(steps))
fetch = sum(v.get("FETCH_SIZE", 0) * per_step(v) for v in step.values())
write = sum(v.get("WRITE_SIZE", 0) * per_step(v) for v in step.values())
main = max(step.items(), key=lambda kv: kv[1].get("avg_ns", 0) * per_step(kv[1]))
m, lps = main[1], per_step(main[1])
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (kernel_source_hash: the code these counters belong to)
out = {"workload": workload, "seeds_per_gpu": seeds, "source_hash": bench.kernel_source_hash(), "source": os.path.basename(os.path.dirname(d) if d.endswith(".json") else d.rstrip("/")),
       "dominant_kernel": main[0][:60], "dominant_kernel_avg_ms": round(m["avg_ns"] / 1e6, 4), "dominant_kernel_launches_per_step": lps,
       "valu_lane_insts_per_cell": round(sum(v.get("SQ_INSTS_VALU", 0) * per_step(v) for k, v in step.items() if "finalize" not in k) * 64 / cells, 2),
       "valu_issue_busy": round(m["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (m["GRBM_GUI_ACTIVE"] / 8), 3),
       "waves_per_simd_avg": round(m["SQ_WAVE_CYCLES"] * 4 / 1024 / (m["GRBM_GUI_ACTIVE"] / 8), 2),
       "clock_ghz": round(m["GRBM_GUI_ACTIVE"] / 8 / m["avg_ns"], 3)}
if fetch or write:
    out.update({"FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write,
                "traffic_kernels": {k[:40]: int((2 * v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)) * 1024 * per_step(v)) for k, v in step.items()}})
path = os.path.join(ROOT, "profiles", "pmc_latest.json")
try:
    allw = json.load(open(path))
    if "workload" in allw:               # round 4's single-entry format
        allw = {allw["workload"]: allw}
except Exception:
    allw = {}
allw[sys.argv[6] if len(sys.argv) > 6 else workload] = out
json.dump(allw, open(path, "w"), indent=1)
print(json.dumps(out))
