# joint-limit rounds of one or two columns as a function of their own (sp2 / sp4: config-2 / config-4 builds) against the product
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r05
bash scripts/ab.sh "product sp2 product sp2" "2" sp
bash scripts/ab.sh "product sp4 product sp4" "4" sp4
for v in product sp2; do if [ $v = product ]; then unset ORC_LIB; else export ORC_LIB=$GRAFT_REPO_ROOT/or_cdchomp_amd/liborcdchomp_var_$v.so; fi; echo "== $v"; timeout -k 10 120 python scripts/phase_profile.py 2>&1 | grep "kernel\|joint limits\|smooth+solve"; done
