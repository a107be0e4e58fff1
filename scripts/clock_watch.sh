#!/bin/bash
# samples the shader clock and power while large batches run (run through gpurun)
cd ${GRAFT_REPO_ROOT:-/root/repo}
python scripts/quick_bench.py ${1:-6144} ${2:-40} > gpurun_out/clock_bench.log 2>&1 &
PID=$!
for i in $(seq 1 120); do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Current Socket" | sed 's/.*(\([0-9]*Mhz\)).*/\1/; s/.*Power (W): //' | tr '\n' ' '; echo
  kill -0 $PID 2>/dev/null || break
  sleep 0.2
done | awk '{c[$0]++} END{for (k in c) print c[k], k}' | sort -k2 -n
wait $PID
tail -1 gpurun_out/clock_bench.log
