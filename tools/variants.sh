#!/bin/bash
# usage: tools/variants.sh OUT name1 name2 ... : headline + mixed bins GCUPS for library variants (libbwasw_<name>.so)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; shift; mkdir -p $OUT
A="--no-e2e --no-cpu-baseline --no-extra --steps 10 --warmup 2"
for v in "$@"; do
  for rep in 1 2; do
    BSW_LIB_PATH=$R/bwa-mem-sw_amd/libbwasw_$v.so python3 $R/bench.py $A > $OUT/${v}_head_$rep.json 2>/dev/null
    BSW_LIB_PATH=$R/bwa-mem-sw_amd/libbwasw_$v.so python3 $R/bench.py $A --workload 150bp_w100_mixed_bins > $OUT/${v}_mixed_$rep.json 2>/dev/null
  done
done
python3 - $OUT <<'PY'
import json, sys, os, glob
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    try:
        j = json.load(open(f)); print(os.path.basename(f), j["value"], j["ms_per_step"])
    except Exception as e: print(f, "ERR", e)
PY
