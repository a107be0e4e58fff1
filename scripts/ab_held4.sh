#!/bin/bash
# A/B of experiment builds on the held4 workload:  scripts/ab_held4.sh "<variants: product x y>" [reps=2]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for rep in $(seq 1 ${2:-2}); do
for v in $1; do
  if [ $v = product ]; then unset ORC_LIB; else export ORC_LIB=$ROOT/or_cdchomp_amd/liborcdchomp_var_$v.so; fi
  timeout -k 10 120 python3 scripts/held4_rate.py 2>&1 | tail -1
done; done
