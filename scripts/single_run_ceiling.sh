#!/bin/bash
# What would ONE run gain from having its tiles on several CUs?  A timing experiment (the results of the 9-mode runs are
# wrong on purpose): ORC_STAGGER_MODE=9 walks only the first tile of each iteration and pauses ORC_STAGGER_SLEEPS*640
# cycles where a barrier across workgroups would be.  Output -> profiles/rNN_single_run_ceiling.txt
cd "$(dirname "$0")/.."
out=${1:-gpurun_out/single_run_ceiling.txt}
: > $out
echo "# baseline (the plan as shipped)" >> $out
ORC_DEBUG_PLAN=1 python scripts/single_run_latency.py 100 2>&1 | grep -E "plan|one WAM" | sort | uniq -c | sort -rn | head -4 >> $out
for threads in 512 256; do
for tm in 49 33 25; do
  echo "# tile_m=$tm threads=$threads G in global, all tiles walked by one workgroup" >> $out
  ORC_BLOCK_THREADS=$threads ORC_TILE_M=$tm ORC_G_LDS=0 python scripts/single_run_latency.py 100 2>&1 | grep "one WAM" >> $out
  for sl in 0 4 8; do
  echo "# tile_m=$tm threads=$threads first tile only + pause of $((sl*640)) cycles" >> $out
  ORC_STAGGER_MODE=9 ORC_STAGGER_SLEEPS=$sl ORC_BLOCK_THREADS=$threads ORC_TILE_M=$tm ORC_G_LDS=0 python scripts/single_run_latency.py 100 2>&1 | grep "one WAM" >> $out
  done
done
done
echo "# where the cycles of the shipped plan go" >> $out
python scripts/phase_profile_single.py 100 2>&1 | grep -v "^orc plan" >> $out
cat $out
