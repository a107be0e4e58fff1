#!/bin/bash
# quick PMC pass: instruction mix of the iterate kernel (run through gpurun)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmcq
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $OUT/a -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/b -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/b.log 2>&1
python3 - <<'PY'
import csv, glob, collections, os
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/pmcq"
for sub in ("a", "b"):
    for f in glob.glob(root + "/%s/*/*counter_collection.csv" % sub):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "chomp_iterate" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()):
            m = sum(v) / len(v)
            print("%-28s %.4g per launch   %.1f per run-iteration" % (k, m, m / (1024 * 101)))
PY
