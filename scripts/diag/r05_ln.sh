# a lean copy of the update phase for runs of the common kind (ln2 / ln4: config-2 / config-4 builds) against the product
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r05
bash scripts/ab.sh "product ln2 product ln2" "2" ln
bash scripts/ab.sh "product ln4 product ln4" "4" ln4
for v in product ln2; do if [ $v = product ]; then unset ORC_LIB; else export ORC_LIB=$GRAFT_REPO_ROOT/or_cdchomp_amd/liborcdchomp_var_$v.so; fi; echo "== $v"; timeout -k 10 120 python scripts/phase_profile.py 2>&1 | grep "kernel\|joint limits\|smooth+solve"; done
