#!/bin/bash
# A variant of libpem_hip.so that differs in ONE translation unit: tools/build_variant_fast.sh <name> <unit (e.g. pem_kernels)> [-DMACRO=1 ...]
# recompiles that unit with the flags and links it with the other units' objects of the regular build (hallthrusterpem_amd/build/).
set -e
cd "$(dirname "$0")/.."
name=$1; unit=$2; shift 2
python -c "from hallthrusterpem_amd import build; build.build()"
mkdir -p build_variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Ihallthrusterpem_amd/csrc "$@" -c hallthrusterpem_amd/csrc/$unit.hip -o build_variants/${unit}_$name.o
objs=$(ls hallthrusterpem_amd/build/*.o | grep -v "/$unit.o")
hipcc --offload-arch=gfx950 -fPIC -shared build_variants/${unit}_$name.o $objs -o build_variants/libpem_$name.so
echo build_variants/libpem_$name.so
