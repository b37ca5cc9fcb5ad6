#!/bin/bash
# A second build of the library with extra compiler flags for kernels.hip, for A/B timing inside one GPU call:
#   tools/build_variant.sh <name> [-DRND_...=...]   ->   build_ab/lib_<name>.so   (use with RND_LIBRARY=build_ab/lib_<name>.so)
set -e
name=$1; shift
cd "$(dirname "$0")/.."
mkdir -p build_ab
C=renderer_amd/csrc
make -C $C -j8 > /dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-function -Xclang -target-feature -Xclang -packed-fp32-ops \
    "$@" -c $C/kernels.hip -o build_ab/kernels_$name.o 2> >(grep -v "is not a recognized feature for this target" >&2)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_ab/lib_$name.so build_ab/kernels_$name.o $C/visual_server.o $C/capi.o $C/comm.o -ldl
rm -f build_ab/kernels_$name.o
echo build_ab/lib_$name.so
