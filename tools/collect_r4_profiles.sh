#!/bin/bash
# gpurun_out/r4prof (written by tools/profile_r4.sh on the GPU box) -> profiles/r4 + profiles/pmc_latest.json
set -e
cd "$(dirname "$0")/.."
for w in head mixed w250 c72 wave quad; do
    d=gpurun_out/r4prof/$w
    ks=$(ls -t $d/trace/*/*_kernel_stats.csv | head -1)
    cp $ks profiles/r4/kernel_stats_$w.csv
    cp $d/summary.json profiles/r4/pmc_summary_$w.json
done
cp gpurun_out/r4prof/head/bench_trace.json profiles/r4/bench_under_rocprof.json
python3 tools/make_pmc_latest.py gpurun_out/r4prof/head/summary.json 150bp_w100_single_bin 1000000 15750334141
