cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r05
bash scripts/test_toggles.sh > gpurun_out/r05/test_toggles.txt 2>&1; echo "toggles rc $?"; grep -c "passed" gpurun_out/r05/test_toggles.txt; grep -c "failed" gpurun_out/r05/test_toggles.txt
