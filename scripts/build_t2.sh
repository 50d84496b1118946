#!/bin/bash
# build_t2.sh: the -DFD_TIMING2 build of the convolution kernels as a variant library (flowdec_amd/variants/libflowdec_t2.so), built
# LOCALLY (hipcc cross-compiles); select it on the GPU box with FLOWDEC_HIP_LIB.
set -e
cd "$(dirname "$0")/.."
mkdir -p flowdec_amd/variants flowdec_amd/build
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Iinclude -Iflowdec_amd/csrc -Wno-unused-result -DFD_TIMING2"
for s in conv_mfma conv_wino conv_wino4; do /opt/rocm/bin/hipcc $F -c flowdec_amd/csrc/$s.hip -o flowdec_amd/build/${s}_t2.o & P="$P $!"; done
for p in $P; do wait $p; done   # (a bare `wait` would swallow a failed compile)
OTHERS=$(ls flowdec_amd/build/{api,calib,conv_head,elementwise,stft,model,ndac,ndac_mfma}.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o flowdec_amd/variants/libflowdec_t2.so flowdec_amd/build/conv_mfma_t2.o flowdec_amd/build/conv_wino_t2.o flowdec_amd/build/conv_wino4_t2.o $OTHERS
echo built flowdec_amd/variants/libflowdec_t2.so
