# TSR lines: lean update phase with constraints + the dense constraint step as its own function (t7) against the product
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r05
AB_EXTRA="--steps 10 --warmup 2 --serial-steps 4" bash scripts/ab.sh "product t7 product t7" "tsr1 tsr3" t7
ORC_LIB=$GRAFT_REPO_ROOT/or_cdchomp_amd/liborcdchomp_var_t7.so timeout -k 10 300 python -m pytest tests/test_gpu_tsr.py -q 2>&1 | tail -n 2
