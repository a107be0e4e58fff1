cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
export ORC_LIB=$GRAFT_REPO_ROOT/or_cdchomp_amd/liborcdchomp_var_p8.so
run() { name=$1; shift
  env "$@" ORC_DEBUG_PLAN=1 timeout -k 10 200 python3 bench.py --config 2 --steps 24 --warmup 2 --serial-steps 0 --no-cpu-baseline --no-other-configs > gpurun_out/r05/c2plan.json 2> gpurun_out/r05/c2plan.err
  python3 - "$name" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r05/c2plan.json").read().strip().splitlines()[-1])
print(sys.argv[1], "value %.3f M" % (d["value"] / 1e6))
PY
  grep "orc plan" gpurun_out/r05/c2plan.err | sort | uniq -c | tail -1
}
run "planner" A=1
run "T in LDS" ORC_T_LDS=1
run "T, G in LDS" ORC_T_LDS=1 ORC_G_LDS=1
run "planner again" A=1
run "tile 34 forced" ORC_T_LDS=1 ORC_G_LDS=1 ORC_TILE_M=34
