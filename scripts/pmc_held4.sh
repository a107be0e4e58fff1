#!/bin/bash
# counters of the held4 launch for experiment builds:  scripts/pmc_held4.sh "<variants: product x y>" [tag]   (WGS_PER_CU passes through)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd $ROOT
TAG=${2:-pmch4}; RUNS=1024; ITER=100
for v in $1; do
  if [ $v = product ]; then unset ORC_LIB; else export ORC_LIB=$ROOT/or_cdchomp_amd/liborcdchomp_var_$v.so; fi
  OUT=$ROOT/gpurun_out/${TAG}_$v; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/a -- python3 scripts/run_held4.py $RUNS $ITER > $OUT/a.log 2>&1
  rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/b -- python3 scripts/run_held4.py $RUNS $ITER > $OUT/b.log 2>&1
  tail -n 1 $OUT/a.log
  python3 - "$v" "$OUT" "$RUNS" "$ITER" <<'PY'
import csv, glob, collections, sys
v, out, runs, it = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
agg = collections.defaultdict(list)
meta = {}
for f in glob.glob(out + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "chomp_iterate" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = {k: r[k] for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size")}
print("  %-10s" % v, {k: round(sum(x) / len(x) / (runs * it)) for k, x in sorted(agg.items())}, meta)
PY
done
