#!/bin/bash
# memory-side counters of the iterate kernel (vector L1 / L2), through gpurun:  scripts/pmc_mem.sh <config> <n_runs> <n_iter> [variant]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd $ROOT
CFG=$1; RUNS=$2; ITER=$3; V=${4:-product}
if [ $V != product ]; then export ORC_LIB=$ROOT/or_cdchomp_amd/liborcdchomp_var_$V.so; fi
OUT=$ROOT/gpurun_out/pmcmem_$V; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/a -- python3 scripts/run_cfg.py $CFG $RUNS $ITER > $OUT/a.log 2>&1
# (a second group with TA_BUSY_avr / GRBM_GUI_ACTIVE made rocprofv3 abort at start-up on this pool and the run hang: left out)
tail -n 2 $OUT/a.log
python3 - "$OUT" "$RUNS" "$ITER" <<'PY'
import csv, glob, collections, sys
out, runs, it = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
agg = collections.defaultdict(list)
for f in glob.glob(out + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "chomp_iterate" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, x in sorted(agg.items()):
    m = sum(x) / len(x)
    print("  %-36s %.4g per launch  %.1f per run-iteration" % (k, m, m / (runs * it)))
PY
