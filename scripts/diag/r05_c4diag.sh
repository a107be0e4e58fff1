# scan rounds of config 4 whose columns hold at most K violated entries each (diagnostic builds d1..d3 count them in the "general loop" field)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r05
for k in 1 2 3; do
  echo "== at most $k per column"; ORC_LIB=$GRAFT_REPO_ROOT/or_cdchomp_amd/liborcdchomp_var_d$k.so timeout -k 10 200 python scripts/phase_profile_cfg.py 4 2>&1 | grep "round kinds\|joint limits"
done
