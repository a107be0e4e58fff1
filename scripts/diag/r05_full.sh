cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
timeout -k 10 1000 python -m pytest tests/ -q -m gpu > gpurun_out/r05/gputests_c.txt 2>&1; echo "tests rc $?"; tail -5 gpurun_out/r05/gputests_c.txt
timeout -k 10 600 python bench.py > gpurun_out/r05/bench_default_c.json 2> gpurun_out/r05/bench_default_c.err; echo "bench rc $?"
tail -c 1200 gpurun_out/r05/bench_default_c.json
