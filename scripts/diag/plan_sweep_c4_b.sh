#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "ORC_WGS=2" "ORC_WGS=2 ORC_TILE_M=99" "ORC_TILE_M=50" "ORC_BLOCK_THREADS=192"; do
  env $cfg ORC_DEBUG_PLAN=1 python3 bench.py --config 4 --no-cpu-baseline --no-other-configs --no-sweep --steps 12 --warmup 2 --serial-steps 4 > gpurun_out/plan_c4.log 2> gpurun_out/plan_c4.err
  python3 - "$cfg" <<'PY'
import json, sys
plan = [l for l in open("gpurun_out/plan_c4.err") if l.startswith("orc plan")]
for l in open("gpurun_out/plan_c4.log"):
    if l.startswith("{"):
        d = json.loads(l); print("%-44s value %.3f M serial %.3f M | %s" % (sys.argv[1] or "(planner)", d["value"]/1e6, d["value_serial"]/1e6, plan[-1].strip()[10:] if plan else ""))
PY
done
