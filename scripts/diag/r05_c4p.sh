cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
for v in $1; do
export ORC_LIB=$GRAFT_REPO_ROOT/or_cdchomp_amd/liborcdchomp_var_$v.so
echo "== $v"; timeout -k 10 300 python3 scripts/phase_profile_cfg.py 4 2>&1 | grep -v "orc placement\|orc plan" | tail -12
done
