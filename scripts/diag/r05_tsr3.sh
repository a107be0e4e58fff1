cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
export ORC_LIB=$GRAFT_REPO_ROOT/or_cdchomp_amd/liborcdchomp_var_t7.so
for c in tsr1 tsr3; do
WG_THREADS=128 timeout -k 10 200 python3 scripts/phase_profile_tsr.py $c 2048 2>&1 | grep -v "orc placement" | tail -8
WG_THREADS=0 timeout -k 10 200 python3 scripts/phase_profile_tsr.py $c 1024 2>&1 | grep -v "orc placement" | tail -8
done
