#!/bin/bash
# -m gpu tests under the switches that select the kernel's fallback paths (run through gpurun)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in ORC_AG_LDS=0 ORC_NO_SCAN_SOLVE=1 ORC_NO_PLACEMENT=1 ORC_NO_JT_SCAN=1 ORC_PCR_FULL=1 ORC_WGS=2 ORC_WGS=1 ORC_PCR_LDS=1 ORC_TILE_M=7 ORC_TILE_M=33 "ORC_WGS=1 ORC_TILE_M=49" ORC_BLOCK_THREADS=192 ORC_LIM_GENERIC=1 "ORC_NO_SCAN_SOLVE=1 ORC_BLOCK_THREADS=192" "ORC_LIM_GENERIC=1 ORC_BLOCK_THREADS=192" ORC_TSR_DENSE=1 ORC_BLOCK_THREADS=512 "ORC_NO_SCAN_SOLVE=1 ORC_BLOCK_THREADS=512" "ORC_LIM_GENERIC=1 ORC_BLOCK_THREADS=512"; do
  echo "== $cfg"; env $cfg python -m pytest tests -m gpu -q -x 2>&1 | tail -1
done
