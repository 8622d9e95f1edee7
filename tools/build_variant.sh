#!/bin/bash
# A/B builds of the library:  tools/build_variant.sh NAME "<extra hipcc flags>"  ->  rtg_slam_amd/_variants/NAME.so
# (git-ignored; travels with gpurun; select with RTGS_LIB_PATH=rtg_slam_amd/_variants/NAME.so)
set -e
cd "$(dirname "$0")/../rtg_slam_amd/csrc"
NAME=$1; EXTRA=$2; D=/tmp/variant_$NAME; mkdir -p $D ../_variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-result"
for f in raster_fwd raster_bwd raster_bin raster_api map_ops map_step icp slam_ops; do
  X=""; [ $f = raster_bwd ] && X="-mllvm -amdgpu-atomic-optimizer-strategy=None"; [ $f = icp ] && X="-ffp-contract=off"; [ $f = slam_ops ] && X="-ffp-contract=off"
  if [ $f = raster_fwd ] || [ $f = raster_bwd ] || [ ! -f $D/$f.o ]; then /opt/rocm/bin/hipcc $FLAGS $X $EXTRA -c $f.hip -o $D/$f.o & fi
done; wait
for f in raster_bin raster_api map_ops map_step icp slam_ops; do [ -f $D/$f.o ] || cp $f.o $D/$f.o; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../_variants/$NAME.so $D/*.o
ls -la ../_variants/$NAME.so
