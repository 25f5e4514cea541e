#!/bin/bash
# Build an experimental variant of libpem_hip.so: tools/build_variant.sh <name> [-DMACRO=1 ...]
# -> build_variants/libpem_<name>.so (git-ignored, but it travels to the GPU box with gpurun)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build_variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude -Ihallthrusterpem_amd/csrc "$@" \
    hallthrusterpem_amd/csrc/pem_kernels.hip hallthrusterpem_amd/csrc/pem_sampler.hip hallthrusterpem_amd/csrc/pem_svd.hip hallthrusterpem_amd/csrc/pem_likelihood.hip hallthrusterpem_amd/csrc/pem_surrogate.hip hallthrusterpem_amd/csrc/pem_fp32.hip hallthrusterpem_amd/csrc/pem_saltelli.hip hallthrusterpem_amd/csrc/pem_latent.hip hallthrusterpem_amd/csrc/pem_quantile.hip hallthrusterpem_amd/csrc/pem_masks.hip -o build_variants/libpem_$name.so
echo build_variants/libpem_$name.so
