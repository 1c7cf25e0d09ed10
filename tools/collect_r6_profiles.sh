#!/bin/bash
# gpurun_out/r6prof (written by tools/profile_r6.sh on the GPU box) -> profiles/r6 + profiles/pmc_latest.json
set -e
cd "$(dirname "$0")/.."
mkdir -p profiles/r6
for w in head single w250 mixed1m group c72 wave quad; do
    d=gpurun_out/r6prof/$w
    [ -d $d ] || continue
    ks=$(ls -t $d/trace/*/*_kernel_stats.csv | head -1)
    cp $ks profiles/r6/kernel_stats_$w.csv
    cp $d/summary.json profiles/r6/pmc_summary_$w.json
    cp $d/bench_trace.json profiles/r6/bench_under_rocprof_$w.json
done
cells() { python3 -c "import json,sys; print(int(json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])['cells_per_step']))" $1; }
python3 tools/make_pmc_latest.py gpurun_out/r6prof/head/summary.json 150bp_w100_mixed_bins 10000000 $(cells gpurun_out/r6prof/head/bench_trace.json) 4 150bp_w100_mixed_bins@10000000
python3 tools/make_pmc_latest.py gpurun_out/r6prof/single/summary.json 150bp_w100_single_bin 1000000 $(cells gpurun_out/r6prof/single/bench_trace.json)
python3 tools/make_pmc_latest.py gpurun_out/r6prof/w250/summary.json 250bp_w500 1000000 $(cells gpurun_out/r6prof/w250/bench_trace.json)
python3 tools/make_pmc_latest.py gpurun_out/r6prof/mixed1m/summary.json 150bp_w100_mixed_bins 1000000 $(cells gpurun_out/r6prof/mixed1m/bench_trace.json)
[ -d gpurun_out/r6prof/group ] && python3 tools/make_pmc_latest.py gpurun_out/r6prof/group/summary.json 150bp_w100_mixed_bins 49152 $(cells gpurun_out/r6prof/group/bench_trace.json) 4 150bp_w100_mixed_bins@49152
