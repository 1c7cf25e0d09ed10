#!/bin/bash
# GCUPS of the headline bin, the mixed PE bins and three narrow single bins (72 / 64 / 40 columns), one JSON line each
# usage: tools/bins_sweep.sh OUTDIR [extra bench args]
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
B="python3 $R/bench.py --no-e2e --no-cpu-baseline --no-extra --steps 10 --warmup 2 $*"
$B > $OUT/head.json
$B --workload 150bp_w100_mixed_bins > $OUT/mixed.json
$B --spec seed_len_min=79 --spec seed_len_max=79 > $OUT/c72.json
$B --spec seed_len_min=87 --spec seed_len_max=87 > $OUT/c64.json
$B --spec seed_len_min=111 --spec seed_len_max=111 > $OUT/c40.json
$B --workload 250bp_w500 > $OUT/w250.json
python3 - $OUT <<'PY'
import json, sys, os
for f in ("head", "mixed", "c72", "c64", "c40", "w250"):
    j = json.load(open(os.path.join(sys.argv[1], f + ".json")))
    print(f, j["value"], j["ms_per_step"], j["config"]["kernel_launches_per_step"])
PY
