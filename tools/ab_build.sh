#!/bin/bash
# A/B builds of the library: tools/ab_build.sh NAME [-DEC3D_...=v ...] compiles csrc/ec3d_kernels.hip with the extra
# flags and links it with the current objects of the other sources into tools/abtmp/libec3d_hip_NAME.so (git-ignored,
# pushed to the GPU box; EC3D_LIB=<that file> selects it; delete tools/abtmp/ when the A/B is settled).
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd); b=$root/eddy_currents_3d_amd/csrc/build; mkdir -p $root/tools/abtmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC -Wall -Wno-unused-function "$@" \
    -c $root/eddy_currents_3d_amd/csrc/ec3d_kernels.hip -o $root/tools/abtmp/kernels_$name.o
objs=$(ls $b/*.o | grep -v ec3d_kernels.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -ldl $root/tools/abtmp/kernels_$name.o $objs -o $root/tools/abtmp/libec3d_hip_$name.so
rm -f $root/tools/abtmp/kernels_$name.o
echo built tools/abtmp/libec3d_hip_$name.so
