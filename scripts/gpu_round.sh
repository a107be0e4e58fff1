#!/bin/bash
# The GPU steps of a round, one parameterised script (it replaces the 64 one-off scripts/diag/r05_*.sh of round 5).
# Started through gpurun from the repository root; everything goes under gpurun_out/<tag>/:
#   gpurun --timeout 1200 -- 'bash scripts/gpu_round.sh r06 tests smoke bench'
#   gpurun --timeout 1200 -- 'bash scripts/gpu_round.sh r06 profiles 2 4 5 tsr1 tsr3 held4'      (rocprofv3 trace + PMC passes per config)
#   gpurun --timeout 1200 -- 'WIDE=2000 bash scripts/gpu_round.sh r06 wide toggles'
#   gpurun --timeout 600  -- 'bash scripts/gpu_round.sh r06 phase 4'                               (per-phase cycle counters of a config)
# Steps: tests | smoke | bench | toggles | wide | profiles <configs...> | phase <configs...>.  A failing step stops the script.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
mode=""
for a in "$@"; do
  case $a in
    tests)   mode=""; timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/tests.txt 2>&1; rc=$?; tail -n 3 $OUT/tests.txt; [ $rc = 0 ] || exit $rc;;
    smoke)   mode=""; timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; rc=$?; tail -n 1 $OUT/smoke.txt | cut -c1-300; [ $rc = 0 ] || exit $rc;;
    bench)   mode=""; timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; rc=$?
             tail -c 1200 $OUT/bench_default.json; echo; cp -f bench_full.json $OUT/bench_full.json 2>/dev/null; [ $rc = 0 ] || exit $rc;;
    toggles) mode=""; timeout -k 10 1100 bash scripts/test_toggles.sh > $OUT/test_toggles.txt 2>&1; rc=$?; grep -c passed $OUT/test_toggles.txt; grep -c failed $OUT/test_toggles.txt; [ $rc = 0 ] || exit $rc;;
    wide)    mode=""; ORC_RANDOM_ROBOTS=${WIDE:-2000} timeout -k 10 1100 python -m pytest tests/test_gpu_random_robots.py -q > $OUT/random_robots_wide.txt 2>&1; rc=$?
             tail -n 1 $OUT/random_robots_wide.txt; [ $rc = 0 ] || exit $rc;;
    profiles|phase) mode=$a;;
    *) if [ "$mode" = profiles ]; then
         timeout -k 10 600 bash scripts/profile_round.sh ${TAG}_$a $a > $OUT/prof_$a.log 2>&1 || exit 1
         tail -c 300 $OUT/prof_$a.log | tr '\n' ' '; echo
       elif [ "$mode" = phase ]; then
         timeout -k 10 300 python3 scripts/phase_profile_cfg.py $a > $OUT/phase_cycles_config$a.txt 2>&1 || exit 1
         grep -v "orc placement\|orc plan" $OUT/phase_cycles_config$a.txt | tail -n 14
       else echo "gpu_round.sh: unknown step $a"; exit 2; fi;;
  esac
done
