cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r05
for cfg in "ORC_LIM_GENERIC=1" "ORC_NO_SCAN_SOLVE=1" "ORC_T_LDS=0 ORC_G_LDS=0" "ORC_BLOCK_THREADS=128"; do
  echo "== $cfg"; env $cfg ORC_RANDOM_ROBOTS=1000 timeout -k 10 600 python -m pytest tests/test_gpu_random_robots.py -q -x 2>&1 | tail -n 1
done > gpurun_out/r05/random_robots_toggles2.txt 2>&1
cat gpurun_out/r05/random_robots_toggles2.txt
