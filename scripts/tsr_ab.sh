#!/bin/bash
# the two TSR bench lines for a list of experiment builds (csrc/Makefile `var`, built with -DORC_TSR_TIMERS or not):
#   scripts/tsr_ab.sh "<variants>" [tag]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
TAG=${2:-tsr}
for v in $1; do
for c in tsr1 tsr3; do
  if [ $v = product ]; then unset ORC_LIB; else export ORC_LIB=$ROOT/or_cdchomp_amd/liborcdchomp_var_$v.so; fi
  timeout -k 10 200 python3 bench.py --config $c --no-cpu-baseline --no-other-configs --no-sweep --steps 3 --warmup 1 --serial-steps 2 > gpurun_out/${TAG}_${v}_$c.log 2>&1
  echo "$v $c rc $?: $(grep 'tsr step' gpurun_out/${TAG}_${v}_$c.log | tail -1)"
  grep "^{" gpurun_out/${TAG}_${v}_$c.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('   value %.3f M  serial %.3f M  kernel %.1f ms  parity %s' % (d['value']/1e6, (d['value_serial'] or 0)/1e6, d['roofline']['avg_kernel_ms'], d.get('parity_rel_l2_max_vs_oracle')))"
done
done
