#!/bin/bash
# tools/ab_build.sh NAME [extra hipcc flags...] -- build an A/B variant of the float backend into
# approximatenn_amd/csrc/ab/NAME/ (built .so files are git-ignored but travel to the GPU box).
# Use:  ANN_HIP_LIBDIR=approximatenn_amd/csrc/ab/NAME python bench.py ...
set -e
cd "$(dirname "$0")/../approximatenn_amd/csrc"
name=$1; shift
mkdir -p ab/$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wall -Wno-unused-function -pthread \
    -DUSE_FLOAT "$@" -o ab/$name/libapproxnn_hip_f32.so ann_host.hip ann_saveio.cpp ann_synth.cpp
if [ -n "$AB_F64" ]; then  # AB_F64=1: the double build too
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wall -Wno-unused-function -pthread \
      "$@" -o ab/$name/libapproxnn_hip_f64.so ann_host.hip ann_saveio.cpp ann_synth.cpp
fi
echo "built ab/$name"
