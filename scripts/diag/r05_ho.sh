# a many-column joint-limit call hands over to the one- / two-column code once only that many columns are open (ho4 / ho2) against not (ho40 / ho20)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r05
bash scripts/ab.sh "ho40 ho4 ho40 ho4" "4" ho4
bash scripts/ab.sh "ho20 ho2 ho20 ho2" "2" ho2
for v in ho40 ho4; do echo "== $v"; ORC_LIB=$GRAFT_REPO_ROOT/or_cdchomp_amd/liborcdchomp_var_$v.so timeout -k 10 200 python scripts/phase_profile_cfg.py 4 2>&1 | grep "config 4:\|joint limits\|round kinds"; done
