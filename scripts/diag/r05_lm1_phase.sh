# phase cycles of config 2 (default shape and 192 threads) under the product and the mask-first joint-limit rounds
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r05
for v in product lm1; do for t in 0 192; do
  if [ $v = product ]; then unset ORC_LIB; else export ORC_LIB=$GRAFT_REPO_ROOT/or_cdchomp_amd/liborcdchomp_var_$v.so; fi
  echo "== $v threads $t"; ORC_BLOCK_THREADS=$t timeout -k 10 120 python scripts/phase_profile.py 2>&1 | head -14
done; done > gpurun_out/r05/lm1_phase_c2.txt 2>&1
cat gpurun_out/r05/lm1_phase_c2.txt
