# a pass's costs completed inside the next update phase (lag) against phase_costs every iteration (lag0): config 2 builds
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r05
bash scripts/ab.sh "lag0 lag lag0 lag" "2" lag
for v in lag0 lag; do echo "== $v"; ORC_LIB=$GRAFT_REPO_ROOT/or_cdchomp_amd/liborcdchomp_var_$v.so timeout -k 10 120 python scripts/phase_profile.py 2>&1 | head -8; done
