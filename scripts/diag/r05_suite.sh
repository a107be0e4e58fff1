cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r05
timeout -k 10 600 python -m pytest tests/ -q -m gpu > gpurun_out/r05/gputests_ws.txt 2>&1; echo "tests rc $?"; tail -n 60 gpurun_out/r05/gputests_ws.txt | cut -c1-250
