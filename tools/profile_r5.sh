#!/bin/bash
# Round 5's rocprofv3 evidence in one go (run through gpurun; the launch chain stays ON under --pmc: its wait is bounded):
# headline with every PMC group, then kernel-trace + SQ counters for PE mixed bins, 250 bp, the 72-column class and the
# general kernels.  -> gpurun_out/r5prof; tools/collect_r5_profiles.sh copies the summaries into profiles/r5 and rebuilds
# profiles/pmc_latest.json (one entry per workload) from them.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
rm -rf gpurun_out/r5prof
tools/profile.sh r5prof/head > /dev/null
echo head done
TRAFFIC=1 tools/profile_quick.sh r5prof/mixed --workload 150bp_w100_mixed_bins > /dev/null
echo mixed done
TRAFFIC=1 tools/profile_quick.sh r5prof/w250 --workload 250bp_w500 > /dev/null
echo w250 done
# the PE leg's batch size (bench.py pe_mixed_bins: resident batches of 32 x 128 Ki seeds)
TRAFFIC=1 tools/profile_quick.sh r5prof/mixed4m --workload 150bp_w100_mixed_bins --tasks 4194304 > /dev/null
echo mixed4m done
if [ -z "$QUICK" ]; then
tools/profile_quick.sh r5prof/c72 --spec seed_len_min=79 --spec seed_len_max=79 > /dev/null
BSW_QUAD=0 tools/profile_quick.sh r5prof/wave --kernel 1 --tasks 131072 > /dev/null
tools/profile_quick.sh r5prof/quad --kernel 1 --tasks 131072 > /dev/null
fi
python3 tools/pmc_summary.py gpurun_out/r5prof/head bsw > gpurun_out/r5prof/head/summary.json
echo profiled
