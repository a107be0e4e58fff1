cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r05
timeout -k 10 200 python scripts/diag/c4_rounds_dump.py 2>&1 | tail -1 && python scripts/diag/c4_rounds_fit.py
