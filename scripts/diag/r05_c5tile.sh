cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
export ORC_LIB=$GRAFT_REPO_ROOT/or_cdchomp_amd/liborcdchomp_var_g5g.so
for t in 0 18 17 19; do
  if [ $t = 0 ]; then unset ORC_TILE_M; else export ORC_TILE_M=$t; fi
  ORC_DEBUG_PLAN=1 timeout -k 10 300 python3 bench.py --config 5 --steps 6 --warmup 1 --serial-steps 2 --no-cpu-baseline > gpurun_out/r05/c5tile.json 2> gpurun_out/r05/c5tile.err
  python3 - "$t" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r05/c5tile.json").read().strip().splitlines()[-1])
print("tile", sys.argv[1], "value %.3f M serial %.3f M parity %.2g" % (d["value"] / 1e6, d["value_serial"] / 1e6, d["parity_rel_l2_max_vs_oracle"]))
PY
  grep "orc plan" gpurun_out/r05/c5tile.err | tail -1
done
