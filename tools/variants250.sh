#!/bin/bash
# usage: tools/variants250.sh name1 name2 ... : 250 bp / w = 500 GCUPS without query Ns for library variants (libbwasw_<name>.so)
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in "$@"; do
  BSW_LIB_PATH=$R/bwa-mem-sw_amd/libbwasw_$v.so python3 $R/bench.py --no-e2e --no-cpu-baseline --no-extra --steps 5 --warmup 2 --workload 250bp_w500 --spec n_rate=0 2>/dev/null | python3 -c "import json,sys; j=json.load(sys.stdin); print('$v', 'n_rate=0', j['value'], j['ms_per_step'])"
  BSW_LIB_PATH=$R/bwa-mem-sw_amd/libbwasw_$v.so python3 $R/bench.py --no-e2e --no-cpu-baseline --no-extra --steps 5 --warmup 2 --workload 250bp_w500 2>/dev/null | python3 -c "import json,sys; j=json.load(sys.stdin); print('$v', 'as benched', j['value'], j['ms_per_step'])"
done
