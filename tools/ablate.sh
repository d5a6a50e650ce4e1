#!/bin/bash
# tools/ablate.sh NAME "-DFLAG ..." -- A/B or diagnostic build of the HIP library with extra macros into pathtracer-rs_amd/libptrs_NAME.so
# (bench it with PTRS_LIB=pathtracer-rs_amd/libptrs_NAME.so).  The build id the library carries covers the extra flags, so counters
# measured with such a build are never taken for the product's.
cd "$(dirname "$0")/.."
python3 -m pathtracer-rs_amd.build --variant "$1" $2
