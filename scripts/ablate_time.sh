#!/bin/bash
# Time and vector-instruction share of the parts of the WAM iteration (config 2 shapes, batch 16384 and
# 1024): the product build against diagnostic builds with one part removed (csrc/Makefile `ablate`;
# wrong results, timing only).  All ablations are taken on top of LIM (no joint-limit rounds), because
# removing a force term changes which runs leave their limits.  Run through gpurun:
#   scripts/ablate_time.sh "LIM LIM+ROT LIM+ROTF LIM+SDF LIM+JT LIM+FKSPH LIM+FKSIN"
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd $ROOT
for v in product $1; do
  if [ $v = product ]; then unset ORC_LIB; else export ORC_LIB=$ROOT/or_cdchomp_amd/liborcdchomp_ablate_$v.so; fi
  echo "== $v"
  python3 scripts/quick_bench.py 16384,1024 6 2>&1 | tail -2
  OUT=$ROOT/gpurun_out/abl_$v; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT -- python3 scripts/quick_bench.py 4096 1 > $OUT/log 2>&1
  python3 - "$v" "$OUT" <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[2] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "chomp_iterate" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("   per run-iteration:", {k: round(sum(v)/len(v)/(4096*101)) for k, v in sorted(agg.items())})
PY
done
