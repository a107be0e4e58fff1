cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r05
ORC_RANDOM_ROBOTS=6000 timeout -k 10 1100 python -m pytest tests/test_gpu_random_robots.py -q -x > gpurun_out/r05/random_robots_6000.txt 2>&1; echo "wide rc $?"; tail -n 1 gpurun_out/r05/random_robots_6000.txt
