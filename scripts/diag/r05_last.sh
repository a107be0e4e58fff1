cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r05
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
timeout -k 10 600 python -m pytest tests/ -q -m gpu > gpurun_out/r05/gputests_final.txt 2>&1; echo "tests rc $?"; tail -n 1 gpurun_out/r05/gputests_final.txt
