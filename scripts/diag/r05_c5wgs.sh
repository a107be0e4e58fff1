cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
export ORC_LIB=$GRAFT_REPO_ROOT/or_cdchomp_amd/liborcdchomp_var_g5j.so
for w in 0 3; do
  if [ $w = 0 ]; then unset ORC_WGS; else export ORC_WGS=$w; fi
  ORC_DEBUG_PLAN=1 timeout -k 10 300 python3 bench.py --config 5 --steps 6 --warmup 1 --serial-steps 2 --no-cpu-baseline > gpurun_out/r05/c5w.json 2> gpurun_out/r05/c5w.err
  python3 - "$w" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r05/c5w.json").read().strip().splitlines()[-1])
print("ORC_WGS", sys.argv[1], "value %.3f M serial %.3f M" % (d["value"] / 1e6, d["value_serial"] / 1e6))
PY
  grep "orc plan" gpurun_out/r05/c5w.err | tail -1
done
