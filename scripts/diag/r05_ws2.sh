cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r05
for v in ws ws ws1 ws2 ws0; do echo "== $v"; ORC_LIB=$GRAFT_REPO_ROOT/or_cdchomp_amd/liborcdchomp_var_$v.so timeout -k 10 200 python scripts/phase_profile_cfg.py 4 2>&1 | grep "config 4:\|round kinds"; done
