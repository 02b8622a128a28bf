#!/bin/bash
# tools/tie_profile.sh -- on the GPU box: phase timestamps of the tie path (debug build with -DANN_TIE_PROFILE in
# /tmp/tiep, loaded through ANN_HIP_LIBDIR)
set -e
cd /root/repo
mkdir -p /tmp/tiep
(cd approximatenn_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -DUSE_FLOAT -DANN_TIE_PROFILE -pthread -o /tmp/tiep/libapproxnn_hip_f32.so ann_host.hip ann_saveio.cpp ann_synth.cpp)
ANN_HIP_LIBDIR=/tmp/tiep ANN_TIE_TS=1 python tools/tie_probe.py ${1:-6006}
