#!/bin/bash
# A second build of libs2f_hip.so with ONE translation unit compiled under extra flags (probe switches), for S2F_LIB=... runs:
#   bash tools/build_probe_lib.sh dwp.hip "-DS2F_DWP_PROBE"      -> spike2former_amd/libs2f_probe.so
cd "$(dirname "$0")/../spike2former_amd/csrc"
make -s -j8 > /dev/null || exit 1
mkdir -p build_probe
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $BASE $2 -c $1 -o build_probe/${1%.hip}.o || exit 1
OBJS=$(ls build/*.o | grep -v "build/${1%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS build_probe/${1%.hip}.o -o ../libs2f_probe.so
