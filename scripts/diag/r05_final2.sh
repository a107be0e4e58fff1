# final build of the round: suite, default bench, phase profiles, rocprofv3 passes of every bench line, default bench again with the build's own counters
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r05
timeout -k 10 600 python -m pytest tests/ -q -m gpu > gpurun_out/r05/gputests_final.txt 2>&1; echo "tests rc $?"; tail -n 2 gpurun_out/r05/gputests_final.txt | cut -c1-200
WGS_PER_CU=0 timeout -k 10 200 python scripts/phase_profile_held4.py > gpurun_out/r05/held4_phase_after.txt 2>&1
timeout -k 10 300 python scripts/phase_profile_cfg.py 4 > gpurun_out/r05/phase_cycles_config4.txt 2>&1
timeout -k 10 300 python scripts/phase_profile_cfg.py 5 > gpurun_out/r05/phase_cycles_config5.txt 2>&1
timeout -k 10 120 python scripts/phase_profile.py > gpurun_out/r05/phase_cycles_config2.txt 2>&1
for c in 2 held4 tsr1 tsr3 4 5; do
  bash scripts/profile_round.sh r05f_$c $c > gpurun_out/prof_r05f_$c.log 2>&1
done
echo profiles done
for c in 2 4 5 held4 tsr1 tsr3; do case $c in 2|4|5) nm=r05_config$c;; *) nm=r05_$c;; esac; python scripts/summarize_profile.py r05f_$c $nm $c > gpurun_out/r05/sum_$c.txt 2>&1 || echo "summarize $c failed"; done
cp profiles/counters_latest.json gpurun_out/r05/counters_latest.json
timeout -k 10 600 python bench.py > gpurun_out/r05/bench_default_final.json 2> gpurun_out/r05/bench_default_final.err; echo "bench rc $?"
tail -c 900 gpurun_out/r05/bench_default_final.json; echo
