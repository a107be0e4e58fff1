# the build with the update loop's opaque entry index: the draw that faulted, the whole suite; then the mask-first joint-limit rounds (lm1):
# the whole suite again under that build, and the A/B on configs 4 and 2
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r05
timeout -k 10 150 python -m pytest tests/test_gpu_random_robots.py -q -x -k "oracle[14]" > gpurun_out/r05/seed14_fixed.txt 2>&1; echo "seed14 rc $?"
timeout -k 10 600 python -m pytest tests/ -q -m gpu > gpurun_out/r05/gputests_fixed.txt 2>&1; echo "tests rc $?"; tail -n 2 gpurun_out/r05/gputests_fixed.txt | cut -c1-300
ORC_LIB=$GRAFT_REPO_ROOT/or_cdchomp_amd/liborcdchomp_var_lm1.so timeout -k 10 600 python -m pytest tests/ -q -m gpu > gpurun_out/r05/gputests_lm1.txt 2>&1; echo "lm1 tests rc $?"; tail -n 2 gpurun_out/r05/gputests_lm1.txt | cut -c1-300
bash scripts/ab.sh "product lm1 product lm1" "4 2" lm1 
