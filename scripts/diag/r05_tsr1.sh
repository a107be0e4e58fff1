cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
export ORC_LIB=$GRAFT_REPO_ROOT/or_cdchomp_amd/liborcdchomp_var_t7.so
for c in tsr1 tsr3; do for th in -1 128; do
  ORC_DEBUG_PLAN=1 timeout -k 10 200 python3 bench.py --config $c --steps 8 --warmup 2 --serial-steps 4 --no-cpu-baseline --workgroup-threads $th > gpurun_out/r05/tsr_$c.$th.json 2> gpurun_out/r05/tsr_$c.$th.err
  python3 - $c $th <<'PY'
import json, sys
c, th = sys.argv[1], sys.argv[2]
d = json.loads(open("gpurun_out/r05/tsr_%s.%s.json" % (c, th)).read().strip().splitlines()[-1])
print(c, "threads", th, "value %.3g M serial %.3g M parity %.2g" % (d["value"] / 1e6, d["value_serial"] / 1e6, d["parity_rel_l2_max_vs_oracle"]), d["config"]["knobs"]["value"])
PY
  grep "orc plan" gpurun_out/r05/tsr_$c.$th.err | tail -1
done; done
