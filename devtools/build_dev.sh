#!/bin/bash
# development build of the library: -DFDR_DEV keeps the FDR_KNN_* knobs (never shipped, never tested against)
# usage: bash devtools/build_dev.sh NAME [extra hipcc flags]  -> devtools/ab/libNAME.so  (FEDRANN_HIP_LIB=... to use it)
# (-DFDR_SHAPE_MASK=bits -DFDR_LH_MASK=16|32|48: compile only some prefilter shapes -- fedrann_hip.hip: FDR_SHAPE_CASE;
#  the sources are snapshot first: hipcc reads them twice, minutes apart, and an edit in between would split the build)
name=${1:-dev}; shift
mkdir -p devtools/ab
snap=$(mktemp -d /tmp/fdr_build_XXXXXX)
mkdir -p $snap/fedrann_amd $snap/include
cp -r fedrann_amd/csrc $snap/fedrann_amd/ && cp include/fedrann_hip.h $snap/include/
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -Wall -Wno-unused-result -Wno-inline-asm \
  -pthread -DFDR_DEV "$@" $snap/fedrann_amd/csrc/fedrann_hip.hip -o devtools/ab/lib$name.so
rc=$?
rm -rf $snap
exit $rc
