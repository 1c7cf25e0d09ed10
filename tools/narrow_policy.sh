#!/bin/bash
# PE mixed bins with the 72-column class folded into the 136-column one (default policy) and forced on beside it (BSW_NARROW_SHARE=0),
# and the short single bins on both kernels: does the per-chunk policy of bsw_batch.hip (narrow_fold) still hold?
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
A="--steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-extra --check 0"
one() { python3 bench.py $A "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['ms_per_step'], d['config']['kernel_launches_per_step'])"; }
echo "mixed default:       $(one --workload 150bp_w100_mixed_bins)"
echo "mixed narrow forced: $(BSW_NARROW_SHARE=0 one --workload 150bp_w100_mixed_bins)"
for sl in 79 87 99 111 135; do
  echo "single bin seed_len $sl (qlen $((150-sl))): 72-col class $(one --spec seed_len_min=$sl --spec seed_len_max=$sl)   136-col class $(BSW_NO_NARROW=1 one --spec seed_len_min=$sl --spec seed_len_max=$sl)"
done
