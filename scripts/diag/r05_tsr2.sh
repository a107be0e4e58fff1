cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
export ORC_LIB=$GRAFT_REPO_ROOT/or_cdchomp_amd/liborcdchomp_var_t7.so
run() { # name, env...
  name=$1; shift
  env "$@" ORC_DEBUG_PLAN=1 timeout -k 10 200 python3 bench.py --config $CFG --steps 8 --warmup 2 --serial-steps 0 --no-cpu-baseline --workgroup-threads 128 > gpurun_out/r05/tsr_x.json 2> gpurun_out/r05/tsr_x.err
  python3 - "$CFG $name" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r05/tsr_x.json").read().strip().splitlines()[-1])
print(sys.argv[1], "value %.3g M parity %.2g" % (d["value"] / 1e6, d["parity_rel_l2_max_vs_oracle"]))
PY
  grep "orc plan" gpurun_out/r05/tsr_x.err | tail -1
}
for CFG in tsr1 tsr3; do
run "8/CU" A=1
run "7/CU" ORC_WGS128=7
run "6/CU" ORC_WGS128=6
run "5/CU" ORC_WGS128=5
run "8/CU G in LDS" ORC_G_LDS=1
done
