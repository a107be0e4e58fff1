cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r05
bash scripts/test_toggles.sh > gpurun_out/r05/test_toggles.txt 2>&1; echo "toggles rc $?"; grep -c "passed" gpurun_out/r05/test_toggles.txt; grep -c "failed" gpurun_out/r05/test_toggles.txt
ORC_RANDOM_ROBOTS=2000 timeout -k 10 900 python -m pytest tests/test_gpu_random_robots.py -q -x > gpurun_out/r05/random_robots_wide.txt 2>&1; echo "wide rc $?"; tail -n 1 gpurun_out/r05/random_robots_wide.txt
