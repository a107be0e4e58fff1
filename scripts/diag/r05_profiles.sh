# rocprofv3 kernel trace + counter passes of the bench lines, one tag per configuration (scripts/profile_round.sh), then the default bench
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for c in 2 held4 tsr1 tsr3 4 5; do
  bash scripts/profile_round.sh r05_$c $c > gpurun_out/prof_r05_$c.log 2>&1
  tail -c 300 gpurun_out/prof_r05_$c.log | tr '\n' ' '; echo
done
