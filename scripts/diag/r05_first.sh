set -e
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r05
scripts/ubench/lds_atomic > gpurun_out/r05/lds_atomic.txt 2>&1
python scripts/phase_profile_held4.py > gpurun_out/r05/held4_phase_before.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05/held4_trace -- python3 scripts/phase_profile_held4.py > gpurun_out/r05/held4_trace.log 2>&1 || true
ORC_PHASE_TIMERS= python bench.py --steps 6 --warmup 2 > gpurun_out/r05/bench_start.json 2> gpurun_out/r05/bench_start.err
tail -c 1500 gpurun_out/r05/bench_start.json
