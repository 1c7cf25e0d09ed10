#!/bin/bash
# occupancy counters of the lane kernels at 1M and 4M seeds per launch (mixed bins, old class table) + headline timing at both sizes
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export BSW_NO_NARROW=1
A="--no-e2e --no-cpu-baseline --no-extra --steps 3 --warmup 1"
for n in 1000000 4000000; do
  python3 $R/bench.py $A --tasks $n > $OUT/head_$n.json 2>/dev/null
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU --output-format csv -d $OUT/pmc_mixed_$n -- python3 $R/bench.py $A --tasks $n --workload 150bp_w100_mixed_bins > /dev/null 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_mixed_$n -- python3 $R/bench.py $A --tasks $n --workload 150bp_w100_mixed_bins > /dev/null 2>&1
done
python3 - $OUT <<'PY'
import csv, glob, sys, os, collections, json
out = sys.argv[1]
for n in (1000000, 4000000):
    j = json.load(open(os.path.join(out, "head_%d.json" % n))); print("head", n, j["value"], j["ms_per_step"])
    agg = collections.defaultdict(list)
    for f in glob.glob(os.path.join(out, "pmc_mixed_%d" % n, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if "lane2" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("mixed", n, {k: sum(v) / len(v) for k, v in agg.items()})
    for f in glob.glob(os.path.join(out, "trace_mixed_%d" % n, "*", "*_kernel_stats.csv")):
        for r in csv.DictReader(open(f)):
            if "lane2" in r["Name"] or "finalize" in r["Name"]:
                print("  ", r["Name"][:40], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
