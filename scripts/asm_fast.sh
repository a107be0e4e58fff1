#!/bin/bash
# device assembly of the config-2 kernels only (ORC_FAST_BUILD), for static instruction counts:  scripts/asm_fast.sh <out.s> [extra -D...]
OUT=$1; shift
cd $(dirname $0)/../or_cdchomp_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DORC_FAST_BUILD=${ORC_FAST:-2} "$@" -x hip --cuda-device-only -S chomp_kernel.hip -o $OUT 2>&1 | grep -v "hip-link" || true
