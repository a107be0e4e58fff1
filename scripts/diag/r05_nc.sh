cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for k in 1 2 3 4; do echo "== calls with at most $k columns"; ORC_LIB=$GRAFT_REPO_ROOT/or_cdchomp_amd/liborcdchomp_var_n$k.so timeout -k 10 200 python scripts/phase_profile_cfg.py 4 2>&1 | grep "round kinds"; done
