#!/bin/bash
# gpurun_out/r5prof (written by tools/profile_r5.sh on the GPU box) -> profiles/r5 + profiles/pmc_latest.json
set -e
cd "$(dirname "$0")/.."
mkdir -p profiles/r5
for w in head mixed w250 mixed4m c72 wave quad; do
    d=gpurun_out/r5prof/$w
    [ -d $d ] || continue
    ks=$(ls -t $d/trace/*/*_kernel_stats.csv | head -1)
    cp $ks profiles/r5/kernel_stats_$w.csv
    cp $d/summary.json profiles/r5/pmc_summary_$w.json
    cp $d/bench_trace.json profiles/r5/bench_under_rocprof_$w.json
done
cells() { python3 -c "import json,sys; print(int(json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])['cells_per_step']))" $1; }
python3 tools/make_pmc_latest.py gpurun_out/r5prof/head/summary.json 150bp_w100_single_bin 1000000 $(cells gpurun_out/r5prof/head/bench_trace.json)
python3 tools/make_pmc_latest.py gpurun_out/r5prof/mixed/summary.json 150bp_w100_mixed_bins 1000000 $(cells gpurun_out/r5prof/mixed/bench_trace.json)
python3 tools/make_pmc_latest.py gpurun_out/r5prof/w250/summary.json 250bp_w500 1000000 $(cells gpurun_out/r5prof/w250/bench_trace.json)
[ -d gpurun_out/r5prof/mixed4m ] && python3 tools/make_pmc_latest.py gpurun_out/r5prof/mixed4m/summary.json 150bp_w100_mixed_bins 4194304 $(cells gpurun_out/r5prof/mixed4m/bench_trace.json) 4 150bp_w100_mixed_bins@4194304
