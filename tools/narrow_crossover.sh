#!/bin/bash
# where does the 72-column class start to pay?  PE workloads with different seed-length ranges, class forced on / off
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; mkdir -p $OUT
A="--no-e2e --no-cpu-baseline --no-extra --steps 8 --warmup 2 --workload 150bp_w100_mixed_bins"
for rng in "19 60" "30 80" "40 100" "50 110" "60 120" "70 130" "80 140"; do
  set -- $rng
  for sh in 0 2; do
    BSW_NARROW_SHARE=$sh python3 $R/bench.py $A --spec seed_len_min=$1 --spec seed_len_max=$2 > $OUT/s$1_$2_share$sh.json 2>/dev/null
  done
done
python3 - $OUT <<'PY'
import json, sys, os, glob
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    try:
        j = json.load(open(f)); print(os.path.basename(f), j["value"], j["ms_per_step"], j["config"]["kernel_launches_per_step"])
    except Exception as e: print(f, "ERR", e)
PY
