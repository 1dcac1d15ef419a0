#!/bin/bash
# development build of the library: -DFDR_DEV keeps the FDR_KNN_* knobs (never shipped, never tested against)
# usage: bash devtools/build_dev.sh NAME [extra hipcc flags]  -> devtools/ab/libNAME.so  (FEDRANN_HIP_LIB=... to use it)
name=${1:-dev}; shift
mkdir -p devtools/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -Wall -Wno-unused-result -Wno-inline-asm \
  -pthread -DFDR_DEV "$@" fedrann_amd/csrc/fedrann_hip.hip -o devtools/ab/lib$name.so
