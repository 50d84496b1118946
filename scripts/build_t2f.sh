#!/bin/bash
# build_t2f.sh: the -DFD_TIMING2 build of the float32 Winograd kernels (conv_wino4f.hip, conv_wino44f.hip) as a variant library
# (flowdec_amd/variants/libflowdec_t2f.so), built LOCALLY after the product build; select it on the GPU box with FLOWDEC_HIP_LIB
# (scripts/wino44f_timing2.py reads the per-workgroup phase stamps).
set -e
cd "$(dirname "$0")/.."
mkdir -p flowdec_amd/variants flowdec_amd/build
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Iinclude -Iflowdec_amd/csrc -Wno-unused-result -DFD_TIMING2"
for s in conv_mfma conv_wino4f conv_wino44f; do /opt/rocm/bin/hipcc $F -c flowdec_amd/csrc/$s.hip -o flowdec_amd/build/${s}_t2f.o & P="$P $!"; done
for p in $P; do wait $p; done   # (a bare `wait` would swallow a failed compile)
OTHERS=$(ls flowdec_amd/build/{api,calib,conv_head,conv_wino,conv_wino4,elementwise,stft,model,ndac,ndac_mfma}.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o flowdec_amd/variants/libflowdec_t2f.so flowdec_amd/build/conv_mfma_t2f.o flowdec_amd/build/conv_wino4f_t2f.o flowdec_amd/build/conv_wino44f_t2f.o $OTHERS
echo built flowdec_amd/variants/libflowdec_t2f.so
