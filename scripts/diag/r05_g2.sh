cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
timeout -k 10 600 python -m pytest tests/test_gpu_grabbed.py tests/test_gpu_commands.py tests/test_gpu_random_robots.py tests/test_gpu_c_client.py -x -q -s 2>&1 | tail -12
bash scripts/ab.sh "f2e f2d f2e f2d" "2" f2d
