cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r05
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r05/smoke_final.txt 2>&1; echo "smoke rc $?"; tail -n 2 gpurun_out/r05/smoke_final.txt | cut -c1-300
ORC_RANDOM_ROBOTS=2000 timeout -k 10 900 python -m pytest tests/test_gpu_random_robots.py -q -x > gpurun_out/r05/random_robots_wide.txt 2>&1; echo "wide rc $?"; tail -n 1 gpurun_out/r05/random_robots_wide.txt
