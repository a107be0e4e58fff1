cd $GRAFT_REPO_ROOT
for t in 0 32 16 33; do
  if [ $t = 0 ]; then unset ORC_TILE_M; else export ORC_TILE_M=$t; fi
  echo "== config 4 tile $t" >> gpurun_out/tile.txt
  ORC_DEBUG_PLAN=1 python3 bench.py --config 4 --no-cpu-baseline --no-other-configs --steps 6 --warmup 1 --serial-steps 3 > gpurun_out/tile_$t.log 2>&1 || exit 1
  grep -m1 "orc plan" gpurun_out/tile_$t.log >> gpurun_out/tile.txt
  python3 -c "
import json,sys
for l in open('gpurun_out/tile_$t.log'):
    if l.startswith('{'):
        d=json.loads(l); print('value %.3f M serial %.3f M' % (d['value']/1e6, (d['value_serial'] or 0)/1e6))
" >> gpurun_out/tile.txt
done
unset ORC_TILE_M
for c in 2 5; do
  echo "== config $c" >> gpurun_out/tile.txt
  ORC_DEBUG_PLAN=1 python3 bench.py --config $c --no-cpu-baseline --no-other-configs --steps 3 --warmup 1 --serial-steps 2 2>&1 | grep "orc plan" | sort | uniq -c >> gpurun_out/tile.txt
done
cat gpurun_out/tile.txt
