#!/bin/bash
# rocprofv3 passes for the round-1 profile (run on the GPU box through gpurun).
# kernel-trace/stats and the PMC passes are separate runs, as the pool requires.
set -x
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$1
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_pmc_sq.log 2>&1
find $OUT -name "*.csv" | head -50
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do echo "== $f"; head -20 $f; done
