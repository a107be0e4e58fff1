cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
timeout -k 10 1100 python -m pytest tests/ -q -m gpu > gpurun_out/r05/gputests_b.txt 2>&1; echo "tests rc $?"; tail -8 gpurun_out/r05/gputests_b.txt
timeout -k 10 300 python scripts/diag/c4_rounds_dump.py 2>&1 | tail -2
