cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
export ORC_LIB=$GRAFT_REPO_ROOT/or_cdchomp_amd/liborcdchomp_var_p8.so
for mode in rot pairs; do
  if [ $mode = pairs ]; then export ORC_PAIRS16=1; else unset ORC_PAIRS16; fi
  ORC_DEBUG_PLAN=1 timeout -k 10 200 python3 bench.py --config 2 --steps 20 --warmup 2 --serial-steps 6 --no-cpu-baseline --no-other-configs > gpurun_out/r05/p16_$mode.json 2> gpurun_out/r05/p16_$mode.err; echo "rc $?"
  python3 - $mode <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r05/p16_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "value %.3f M serial %.3f M parity %.2g outside %d" % (d["value"] / 1e6, d["value_serial"] / 1e6, d["parity_rel_l2_max_vs_oracle"], d["runs_outside_joint_limits"]), d["config"]["knobs"])
PY
  grep "orc plan\|pair list" gpurun_out/r05/p16_$mode.err | sort | uniq -c | tail -4
done
