#!/bin/bash
# NS1 (north_star: "LDS-staged voxel tiles" for the SDF lookup), closed by measurement.
# Upper bound of what ANY staging of field cells in LDS could gain: a build in which the four cell
# reads of every lookup go to LDS (ORC_ABLATE_SDFLDS: the tile's position buffer stands in for the
# staged field -- wrong values, identical instruction stream, no staging cost at all) against the
# product build, on config 2 (119 KB fp64 field, L2 resident) and config 5 (four fp32 fields, 3.2 MB).
# Both builds run with obs_factor 0: the lookups are made and consumed as always, but the (wrong)
# values of the stand-in cannot steer the trajectories, so both builds iterate the same trajectories.
# Also the wave-cycle counters of both builds.  Run through gpurun; needs the ablation library:
#   hipcc -DORC_ABLATE_SDFLDS ... -> or_cdchomp_amd/liborcdchomp_ablate_SDFLDS.so
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
export OBS_FACTOR=0
ABL=$PWD/or_cdchomp_amd/liborcdchomp_ablate_SDFLDS.so
for lib in product lds; do
  if [ $lib = lds ]; then export ORC_LIB=$ABL; fi
  echo "== $lib build"
  python scripts/quick_bench.py 1024,16384 6 2>&1 | tail -2
  NSTREAMS=3 python scripts/quick_bench.py 1024 12 2>&1 | tail -1
  python scripts/phase_profile_cfg.py 5 2>&1 | grep -E "config 5|cost  "
  OUT=gpurun_out/ns1_$lib; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY --output-format csv -d $OUT -- python3 scripts/quick_bench.py 16384 3 > $OUT/log 2>&1
  python3 - $OUT <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "chomp_iterate" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("counters per launch (batch 16384, 100 iterations):", {k: "%.4g" % (sum(v)/len(v)) for k, v in sorted(agg.items())})
PY
done
