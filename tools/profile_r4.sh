#!/bin/bash
# Round 4's rocprofv3 evidence in one go (run through gpurun): headline with every PMC group (-> profiles/pmc_latest.json),
# then kernel-trace + SQ counters for mixed bins, 250 bp, the 72-column class and the two general kernels.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
rm -rf gpurun_out/r4prof
tools/profile.sh r4prof/head > /dev/null
tools/profile_quick.sh r4prof/mixed --workload 150bp_w100_mixed_bins > /dev/null
tools/profile_quick.sh r4prof/w250 --workload 250bp_w500 > /dev/null
tools/profile_quick.sh r4prof/c72 --spec seed_len_min=79 --spec seed_len_max=79 > /dev/null
BSW_QUAD=0 tools/profile_quick.sh r4prof/wave --kernel 1 --tasks 131072 > /dev/null
tools/profile_quick.sh r4prof/quad --kernel 1 --tasks 131072 > /dev/null
python3 tools/pmc_summary.py gpurun_out/r4prof/head bsw > gpurun_out/r4prof/head/summary.json
echo profiled
