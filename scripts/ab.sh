#!/bin/bash
# A/B of experiment builds of the kernel (csrc/Makefile `var`) on the bench workloads, through gpurun:
#   scripts/ab.sh "<variants: product inl1 ...>" "<configs: 2 4 5>" [tag]
# Prints value (two overlapping launches), value_serial and the kernel durations of every (variant, config).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
TAG=${3:-ab}
OUT=$ROOT/gpurun_out/ab_$TAG.txt
: > $OUT
for c in $2; do
  case $c in 2) ARGS="--steps 20 --warmup 2 --serial-steps 6";; 4) ARGS="--steps 6 --warmup 1 --serial-steps 3";; 5) ARGS="--steps 4 --warmup 1 --serial-steps 2";; *) ARGS="";; esac
  for v in $1; do
    if [ $v = product ]; then unset ORC_LIB; else export ORC_LIB=$ROOT/or_cdchomp_amd/liborcdchomp_var_$v.so; fi
    python3 bench.py --config $c --no-cpu-baseline --no-other-configs $ARGS $AB_EXTRA > $ROOT/gpurun_out/ab_${TAG}_${v}_c$c.log 2>&1
    rc=$?
    python3 - "$v" "$c" "$rc" "$ROOT/gpurun_out/ab_${TAG}_${v}_c$c.log" <<'PY' | tee -a $OUT
import json, sys
v, c, rc, path = sys.argv[1:5]
line = None
for l in open(path):
    if l.startswith("{"):
        line = json.loads(l)
if line is None:
    print("%-10s config %s rc %s: no bench line; tail: %s" % (v, c, rc, open(path).read()[-400:].replace("\n", " | ")))
else:
    r = line["roofline"]
    print("%-10s config %s rc %s: value %.3f M  serial %.3f M  kernel %.2f ms (serial %.2f ms)  parity %.2e" % (
        v, c, rc, line["value"] / 1e6, (line["value_serial"] or 0) / 1e6, r["avg_kernel_ms"], r.get("serial_avg_kernel_ms") or 0,
        line["parity_rel_l2_max_vs_oracle"] or -1))
PY
  done
done
