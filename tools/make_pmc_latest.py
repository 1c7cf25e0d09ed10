#!/usr/bin/env python3
"""profiles/pmc_latest.json: the counters bench.py quotes in its `roofline` object, from a tools/profile.sh output
directory (separate rocprofv3 --pmc passes).  HBM bytes per MI355X_MICROARCH.md's gfx950
correction: FETCH_SIZE (KiB) x 2 + WRITE_SIZE (KiB), summed over the DP kernels of one bsw_run step."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = sys.argv[1]
workload, seeds, cells = sys.argv[2], int(sys.argv[3]), int(float(sys.argv[4]))
if d.endswith(".json"):                  # a kept tools/pmc_summary.py output instead of the raw rocprofv3 directories
    allk = json.load(open(d))
else:
    allk = json.loads(subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), d, "bsw"]))
step = {k: v for k, v in allk.items() if any(x in k for x in ("lane2_kernel", "lane_kernel", "pair_finalize", "wave_kernel"))}
fetch = sum(v.get("FETCH_SIZE", 0) for v in step.values())
write = sum(v.get("WRITE_SIZE", 0) for v in step.values())
main = max(step.items(), key=lambda kv: kv[1].get("avg_ns", 0))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (kernel_source_hash: the code these counters belong to)
out = {"workload": workload, "seeds_per_gpu": seeds, "source_hash": bench.kernel_source_hash(), "source": os.path.basename(os.path.dirname(d) if d.endswith(".json") else d.rstrip("/")),
       "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write,
       "traffic_kernels": {k[:40]: int((2 * v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)) * 1024) for k, v in step.items()},
       "dominant_kernel": main[0][:60], "dominant_kernel_avg_ms": round(main[1]["avg_ns"] / 1e6, 4),
       "valu_lane_insts_per_cell": round(main[1]["SQ_INSTS_VALU"] * 64 / cells, 2),
       "valu_issue_busy": round(main[1]["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (main[1]["GRBM_GUI_ACTIVE"] / 8), 3),
       "waves_per_simd_avg": round(main[1]["SQ_WAVE_CYCLES"] * 4 / 1024 / (main[1]["GRBM_GUI_ACTIVE"] / 8), 2),
       "clock_ghz": round(main[1]["GRBM_GUI_ACTIVE"] / 8 / main[1]["avg_ns"], 3)}
json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_latest.json"), "w"), indent=1)
print(json.dumps(out))
