#!/bin/bash
# PMC pass: instruction-cache behaviour of the iterate kernel (run through gpurun)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmci
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQC_TC_INST_REQ --output-format csv -d $OUT/a -- python3 scripts/quick_bench.py 1024 2 > $OUT/a.log 2>&1
python3 - <<'PY'
import csv, glob, collections, os
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/pmci"
for f in glob.glob(root + "/a/*/*counter_collection.csv"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "chomp_iterate" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        m = sum(v) / len(v)
        print("%-28s %.4g per launch   %.1f per run-iteration" % (k, m, m / (1024 * 101)))
PY
tail -3 $OUT/a.log
