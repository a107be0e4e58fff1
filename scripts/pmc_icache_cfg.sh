#!/bin/bash
# instruction-cache counters of the iterate kernel for a BASELINE configuration, through gpurun:
#   scripts/pmc_icache_cfg.sh <config> <n_runs> <n_iter>
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmci_$1
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQC_TC_INST_REQ --output-format csv -d $OUT/a -- python3 scripts/run_cfg.py $1 $2 $3 > $OUT/a.log 2>&1
python3 - "$OUT" "$2" "$3" <<'PY'
import csv, glob, collections, sys
out, runs, it = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
for f in glob.glob(out + "/a/*/*counter_collection.csv"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "chomp_iterate" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        m = sum(v) / len(v)
        print("%-28s %.4g per launch   %.1f per run-iteration" % (k, m, m / (runs * it)))
PY
tail -n 1 $OUT/a.log
