#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in ORC_PCR_LDS=1 ORC_TILE_M=33 "ORC_WGS=1 ORC_TILE_M=49" ORC_G_LDS=0 "ORC_T_LDS=0 ORC_G_LDS=0" ORC_HMC_PLAN_SYNC=1; do
  echo "== $cfg"; env $cfg python -m pytest tests -m gpu -q -x 2>&1 | grep -E "^(FAILED|E  )" | head -8
done
