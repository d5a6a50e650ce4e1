#!/bin/bash
# tools/ablate.sh NAME "-DFLAG ..." -- diagnostic build of the HIP library with ablation macros into pathtracer-rs_amd/libptrs_NAME.so
# (never the product: such a build computes wrong values; bench it with PTRS_LIB=pathtracer-rs_amd/libptrs_NAME.so)
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-function -Wno-unused-parameter $2 -o pathtracer-rs_amd/libptrs_$1.so pathtracer-rs_amd/csrc/ptrs_hip.hip
