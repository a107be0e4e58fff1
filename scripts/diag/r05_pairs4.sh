cd $GRAFT_REPO_ROOT
bash scripts/ab_held4.sh "p6a p6n" 1
export ORC_LIB=$GRAFT_REPO_ROOT/or_cdchomp_amd/liborcdchomp_var_p6a.so
echo "T in LDS:"; ORC_T_LDS=1 timeout -k 10 120 python3 scripts/held4_rate.py 2>&1 | tail -1
echo "T, G in LDS:"; ORC_T_LDS=1 ORC_G_LDS=1 timeout -k 10 120 python3 scripts/held4_rate.py 2>&1 | tail -1
echo "G in LDS:"; ORC_G_LDS=1 timeout -k 10 120 python3 scripts/held4_rate.py 2>&1 | tail -1
ORC_T_LDS=1 ORC_G_LDS=1 ORC_DEBUG_PLAN=1 WGS_PER_CU=4 timeout -k 10 120 python3 scripts/run_held4.py 2>&1 | grep "orc plan" | tail -1
