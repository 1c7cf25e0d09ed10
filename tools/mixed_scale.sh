#!/bin/bash
# mixed PE bins at 1M and 4M seeds per launch, three configurations: packing (rounds per launch) vs kernel rate
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; mkdir -p $OUT
B="python3 $R/bench.py --no-e2e --no-cpu-baseline --no-extra --steps 6 --warmup 2 --workload 150bp_w100_mixed_bins"
for n in 1000000 4000000; do
  BSW_NO_NARROW=1 $B --tasks $n > $OUT/nonarrow_$n.json 2>/dev/null
  BSW_NO_FORK=1 $B --tasks $n > $OUT/narrow_nofork_$n.json 2>/dev/null
  $B --tasks $n > $OUT/narrow_fork_$n.json 2>/dev/null
done
# single bins, old kernel vs new for the narrow ones
for q in 131 111 91 71 51 31 15; do
  s=$((150-q))
  BSW_NO_NARROW=1 python3 $R/bench.py --no-e2e --no-cpu-baseline --no-extra --steps 6 --warmup 2 --spec seed_len_min=$s --spec seed_len_max=$s > $OUT/q${q}_old.json 2>/dev/null
  python3 $R/bench.py --no-e2e --no-cpu-baseline --no-extra --steps 6 --warmup 2 --spec seed_len_min=$s --spec seed_len_max=$s > $OUT/q${q}_new.json 2>/dev/null
done
python3 - $OUT <<'PY'
import json, sys, os, glob
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    try:
        j = json.load(open(f)); print(os.path.basename(f), j["value"], j["ms_per_step"], j["cells_per_step"])
    except Exception as e: print(f, "ERR", e)
PY
