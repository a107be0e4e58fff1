cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r05
export ORC_LIB=$GRAFT_REPO_ROOT/or_cdchomp_amd/liborcdchomp_var_p6t.so
WGS_PER_CU=4 timeout -k 10 200 python scripts/run_held4.py 2>&1 | tail -3
WGS_PER_CU=0 timeout -k 10 200 python scripts/run_held4.py 2>&1 | tail -3
