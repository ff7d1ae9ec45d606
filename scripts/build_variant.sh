#!/bin/bash
# A/B aid: build a variant of the library with extra compiler flags into ./ab_<name>.so (git-ignored; travels with gpurun).
#   bash scripts/build_variant.sh <name> "<extra flags>"
set -e
cd "$(dirname "$0")/../ldpc_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-parameter \
    -mllvm -amdgpu-atomic-optimizer-strategy=None $2 -shared -o ../../ab_$1.so kernels.hip capi.cpp host_codes.cpp host_json.cpp
