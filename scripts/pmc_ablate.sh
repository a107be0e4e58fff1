#!/bin/bash
# VALU instruction count of the iterate kernel for the ablation builds (run through gpurun)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd $ROOT
for v in amd ablate_ROT ablate_SDF ablate_JT; do
  OUT=$ROOT/gpurun_out/pmca_$v; rm -rf $OUT; mkdir -p $OUT
  export ORC_LIB=$ROOT/or_cdchomp_amd/liborcdchomp_$v.so
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
