#!/bin/bash
# -m gpu tests under the switches that select the kernel's fallback paths (run through gpurun)
# test_toggles.sh [first] [count]: the switches first .. first + count - 1 of the list (a gpurun call is at most 20 minutes: 0 18, then 18 18)
cd ${GRAFT_REPO_ROOT:-/root/repo}
FIRST=${1:-0}; COUNT=${2:-1000}; IDX=0
for cfg in ORC_AG_LDS=0 ORC_NO_SCAN_SOLVE=1 ORC_NO_PLACEMENT=1 ORC_NO_JT_SCAN=1 ORC_PCR_FULL=1 ORC_WGS=2 ORC_WGS=1 ORC_PCR_LDS=1 ORC_TILE_M=7 ORC_TILE_M=33 "ORC_WGS=1 ORC_TILE_M=49" ORC_BLOCK_THREADS=192 ORC_LIM_GENERIC=1 "ORC_NO_SCAN_SOLVE=1 ORC_BLOCK_THREADS=192" "ORC_LIM_GENERIC=1 ORC_BLOCK_THREADS=192" ORC_TSR_DENSE=1 ORC_NO_KIND=1 ORC_NO_STATIC_LANES=1 ORC_BLOCK_THREADS=512 "ORC_NO_SCAN_SOLVE=1 ORC_BLOCK_THREADS=512" "ORC_LIM_GENERIC=1 ORC_BLOCK_THREADS=512" ORC_T_LDS=0 "ORC_T_LDS=0 ORC_T_STAGED=0" ORC_G_LDS=0 "ORC_T_LDS=0 ORC_G_LDS=0" ORC_NO_FK_SPLIT=1 ORC_HMC_PLAN_SYNC=1 ORC_NO_PAIRS=1 ORC_BLOCK_THREADS=128 "ORC_T_LDS=1 ORC_G_LDS=1" ORC_NO_SEMISEP=1 ORC_NO_BAND_TOEPLITZ=1 ORC_PAIRS_CHAIN64_ONLY=1 "ORC_NO_SEMISEP=1 ORC_LIM_GENERIC=1"; do
  IDX=$((IDX+1)); if [ $IDX -le $FIRST ] || [ $IDX -gt $((FIRST+COUNT)) ]; then continue; fi
  echo "== $cfg"; env $cfg python -m pytest tests -m gpu -q 2>&1 | tail -1
done
