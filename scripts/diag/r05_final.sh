# the round's final build: the whole -m gpu suite, the default bench (the driver's command), phase profiles, rocprofv3 passes of every bench line,
# then the seeded random robots far beyond the suite's 24 draws
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r05
timeout -k 10 600 python -m pytest tests/ -q -m gpu > gpurun_out/r05/gputests_final.txt 2>&1; echo "tests rc $?"; tail -n 2 gpurun_out/r05/gputests_final.txt | cut -c1-200
timeout -k 10 600 python bench.py > gpurun_out/r05/bench_default_final.json 2> gpurun_out/r05/bench_default_final.err; echo "bench rc $?"
tail -c 900 gpurun_out/r05/bench_default_final.json; echo
WGS_PER_CU=0 timeout -k 10 200 python scripts/phase_profile_held4.py > gpurun_out/r05/held4_phase_after.txt 2>&1
timeout -k 10 300 python scripts/phase_profile_cfg.py 4 > gpurun_out/r05/phase_cycles_config4.txt 2>&1
timeout -k 10 300 python scripts/phase_profile_cfg.py 5 > gpurun_out/r05/phase_cycles_config5.txt 2>&1
for c in 2 held4 tsr1 tsr3 4 5; do
  bash scripts/profile_round.sh r05f_$c $c > gpurun_out/prof_r05f_$c.log 2>&1
done
echo profiles done
ORC_RANDOM_ROBOTS=${WIDE:-2000} timeout -k 10 900 python -m pytest tests/test_gpu_random_robots.py -q -x > gpurun_out/r05/random_robots_wide.txt 2>&1; echo "wide rc $?"; tail -n 3 gpurun_out/r05/random_robots_wide.txt | cut -c1-300
