#!/bin/bash
# Build zk-nullifier-sig_amd/libplume_hip_<name>.so from the current sources with extra flags for plume_kernels.hip only (the other objects are reused
# from the default build when the flags do not concern them):   tests/gpu_debug/build_variant.sh gw20 "-DPLUME_GW=20" [all]
# A third argument "all" recompiles every translation unit with the flags (flags that change shared headers' layout, e.g. -DPLUME_GW=20).
set -e
name=$1; flags=$2; all=$3
root=$(cd "$(dirname "$0")/../.." && pwd)
src=$root/zk-nullifier-sig_amd/csrc
obj=/tmp/plume_variant_$name
mkdir -p $obj
HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function"
if [ "$all" = "all" ]; then
  make -s -C $src -j3 EXTRA="$flags" OUT=../libplume_hip_$name.so OBJ=$obj
else
  /opt/rocm/bin/hipcc $flags $HIPFLAGS -c $src/plume_kernels.hip -o $obj/plume_kernels.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/zk-nullifier-sig_amd/libplume_hip_$name.so $obj/plume_kernels.o $src/plume_agg_kernels.o $src/plume_capi.o
fi
echo built libplume_hip_$name.so
