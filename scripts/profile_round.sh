#!/bin/bash
# rocprofv3 passes over bench.py for one BASELINE configuration (run on the GPU box through gpurun):
#   scripts/profile_round.sh <tag> [config=2] [extra bench args]
# kernel-trace/stats and every PMC pass are separate runs, as the pool requires.  Summaries for
# profiles/ are written by scripts/summarize_profile.py <tag> <name> <config>.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; CFG=${2:-2}; shift; shift
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
python3 -c "from or_cdchomp_amd import _capi; print(_capi.csrc_hash())" > $OUT/csrc_hash.txt
B="bench.py --config $CFG --no-cpu-baseline --serial-steps 0 $@"
ORC_DEBUG_PLAN=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $B --steps 20 --warmup 3 > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $B --steps 2 --warmup 1 > $OUT/bench_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $B --steps 2 --warmup 1 > $OUT/bench_pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 $B --steps 2 --warmup 1 > $OUT/bench_pmc_sq.log 2>&1
tail -n 1 $OUT/bench_trace.log | head -c 600; echo
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do echo "== $f"; head -8 $f; done
