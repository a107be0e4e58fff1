#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "0 0" "1 6" "1 12" "1 20" "2 6" "2 12" "2 20"; do set -- $cfg
  ORC_STAGGER_MODE=$1 ORC_STAGGER_SLEEPS=$2 python3 bench.py --config 2 --no-cpu-baseline --no-other-configs --no-sweep --steps 10 --warmup 2 --serial-steps 10 > gpurun_out/stag_$1_$2.log 2>&1
  python3 - "$1" "$2" gpurun_out/stag_$1_$2.log <<'PY'
import json, sys
for l in open(sys.argv[3]):
    if l.startswith("{"):
        d = json.loads(l); print("stagger mode %s sleeps %s: value %.3f M serial %.3f M" % (sys.argv[1], sys.argv[2], d["value"]/1e6, d["value_serial"]/1e6))
PY
done
