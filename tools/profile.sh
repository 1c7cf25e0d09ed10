#!/bin/bash
# Collect rocprofv3 evidence for the bench kernel on the GPU box (run through gpurun).
# kernel-trace/stats and every PMC group are separate passes, as MI355X_MICROARCH.md prescribes.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-prof}
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-extra ${BENCH_ARGS:-}"
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $ARGS > $OUT/bench_trace.json
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py $ARGS > /dev/null
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py $ARGS > /dev/null
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py $ARGS > /dev/null
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/pmc_lds -- python3 $R/bench.py $ARGS > /dev/null
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_grbm -- python3 $R/bench.py $ARGS > /dev/null
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU --output-format csv -d $OUT/pmc_icache -- python3 $R/bench.py $ARGS > /dev/null
find $OUT -name "*.csv" | head -50
