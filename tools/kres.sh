#!/bin/bash
# Kernel resource usage (VGPRs, scratch, occupancy, LDS) of one .hip file of rtg_slam_amd/csrc:  tools/kres.sh raster_fwd [filter]
cd "$(dirname "$0")/../rtg_slam_amd/csrc"
EXTRA=""
case "$1" in raster_bwd|raster_bwd_entry) EXTRA="-mllvm -amdgpu-atomic-optimizer-strategy=None";; icp|slam_ops) EXTRA="-ffp-contract=off";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics $EXTRA -Rpass-analysis=kernel-resource-usage -c $1.hip -o /tmp/kres_$$.o 2>&1 |
  grep -E "error|Function Name|VGPRs:|Occupancy|LDS Size|ScratchSize" | sed -E 's/\[-Rpass[^]]*\]//g; s/.*remark: +//' | paste - - - - - |
  sed -E 's/Function Name: _ZN4rtgs[0-9]*([A-Za-z0-9_]*kernel[A-Za-z0-9_]*)[^\t]*/\1/' | grep -i "${2:-.}"
rm -f /tmp/kres_$$.o
