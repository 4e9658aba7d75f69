#!/bin/bash
# phase clocks of the wave-bucket D-FPS kernel (debug library built here with -DSA_FPSB_PROF)
cd $GRAFT_REPO_ROOT/3dssd_amd/csrc
mkdir -p /tmp/fpsb
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-honor-nans -fPIC -DSA_FPSB_PROF -shared -o /tmp/fpsb/libfpsb_prof.so fps_bucket.hip
cd $GRAFT_REPO_ROOT
SA3D_LIB=/tmp/fpsb/libfpsb_prof.so python tools/fps_bucket_prof.py
