#!/bin/bash
# kernel-trace + the SQ issue counters only (three passes) for one bench workload: tools/profile_quick.sh <outdir> <bench args...>
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; shift
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-extra --check 0 $*"
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $ARGS > $OUT/bench_trace.json
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py $ARGS > /dev/null
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_grbm -- python3 $R/bench.py $ARGS > /dev/null
if [ -n "$TRAFFIC" ]; then      # HBM bytes too (separate passes, as MI355X_MICROARCH.md prescribes)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py $ARGS > /dev/null
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py $ARGS > /dev/null
fi
python3 $R/tools/pmc_summary.py $OUT bsw > $OUT/summary.json
