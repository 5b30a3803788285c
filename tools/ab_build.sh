#!/bin/bash
# build a variant of libpwpp_hip.so for an A/B run: tools/ab_build.sh <name> [-DFLAG=...]  ->  ab/<name>.so  (PWPP_LIB_PATH selects it)
name=$1; shift
cd "$(dirname "$0")/../patchwork-plusplus_amd" && mkdir -p ../ab && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -Wall -Wno-unused-function "$@" -shared -Wl,--version-script=csrc/pwpp.map -o ../ab/$name.so csrc/pwpp_kernels.hip csrc/pwpp_fit.hip csrc/pwpp_capi.cpp 2>&1 | grep -E "error" ; ls -la ../ab/$name.so
