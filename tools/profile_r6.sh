#!/bin/bash
# Round 6's rocprofv3 evidence in one go (run through gpurun).  bench.py's timed step now includes the staging kernels
# (bsw_run_staged: pack + bin + DP) and its default workload is the metric's: 150 bp PE mixed bins, 10 M seeds.
#   head     the default command (10 M PE seeds, 3 resident batches): kernel trace + every PMC group
#   single   configs[1]: 150 bp single bin, 1 M seeds            w250   250 bp, w = 500, 1 M seeds
#   mixed1m  PE mixed bins at 1 M seeds                          c72 / wave / quad: the other kernels (unless QUICK)
# -> gpurun_out/r6prof; tools/collect_r6_profiles.sh copies the summaries into profiles/r6 and rebuilds profiles/pmc_latest.json.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
rm -rf gpurun_out/r6prof
export BENCH_EXTRA="--kernels-only-steps 0"
BENCH_ARGS="$BENCH_EXTRA" tools/profile.sh r6prof/head > /dev/null
echo head done
TRAFFIC=1 tools/profile_quick.sh r6prof/single $BENCH_EXTRA --workload 150bp_w100_single_bin --tasks 1000000 > /dev/null
echo single done
TRAFFIC=1 tools/profile_quick.sh r6prof/w250 $BENCH_EXTRA --workload 250bp_w500 --tasks 1000000 > /dev/null
echo w250 done
TRAFFIC=1 tools/profile_quick.sh r6prof/mixed1m $BENCH_EXTRA --workload 150bp_w100_mixed_bins --tasks 1000000 > /dev/null
echo mixed1m done
# the kernel between the general ones and the lane kernels: 49 152 PE seeds take bsw_lane2g_kernel under BSW_KERNEL_AUTO
tools/profile_quick.sh r6prof/group $BENCH_EXTRA --workload 150bp_w100_mixed_bins --tasks 49152 > /dev/null
echo group done
if [ -z "$QUICK" ]; then
tools/profile_quick.sh r6prof/c72 $BENCH_EXTRA --workload 150bp_w100_single_bin --tasks 1000000 --spec seed_len_min=79 --spec seed_len_max=79 > /dev/null
BSW_QUAD=0 tools/profile_quick.sh r6prof/wave $BENCH_EXTRA --workload 150bp_w100_single_bin --kernel 1 --tasks 131072 > /dev/null
tools/profile_quick.sh r6prof/quad $BENCH_EXTRA --workload 150bp_w100_single_bin --kernel 1 --tasks 131072 > /dev/null
fi
python3 tools/pmc_summary.py gpurun_out/r6prof/head bsw > gpurun_out/r6prof/head/summary.json
echo profiled
