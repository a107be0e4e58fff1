#!/bin/bash
# VALU/LDS instruction counts of the iterate kernel with and without an environment switch
# usage: pmc_env.sh VAR=VALUE   (run through gpurun)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd $ROOT
for v in base "$1"; do
  OUT=$ROOT/gpurun_out/pmce_$(echo $v | tr '=' '_'); rm -rf $OUT; mkdir -p $OUT
  if [ "$v" != base ]; then export "$v"; fi
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $OUT -- python3 scripts/quick_bench.py 1024 1 > $OUT/log 2>&1
  python3 - "$v" "$OUT" <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[2] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "chomp_iterate" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[1], {k: round(sum(v)/len(v)/(1024*101)) for k, v in sorted(agg.items())})
PY
done
