cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
export ORC_LIB=$GRAFT_REPO_ROOT/or_cdchomp_amd/liborcdchomp_var_g5t.so
timeout -k 10 300 python3 scripts/run_cfg.py 5 4096 100 2>&1 | tail -3
