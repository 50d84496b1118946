#!/bin/bash
# build_w4var.sh NAME FLAGS...: a -DFD_TIMING2 variant library with conv_wino4.hip compiled with extra flags (timing experiments;
# the W4_EXP_* / W4_NRING switches exist after `patch -p0 < scripts/conv_wino4_ablation.patch`)
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Iinclude -Iflowdec_amd/csrc -Wno-unused-result -DFD_TIMING2"
/opt/rocm/bin/hipcc $F "$@" -c flowdec_amd/csrc/conv_wino4.hip -o flowdec_amd/build/conv_wino4_$NAME.o
OTHERS=$(ls flowdec_amd/build/{api,calib,conv_head,elementwise,stft,model,ndac,ndac_mfma}.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o flowdec_amd/variants/libflowdec_$NAME.so flowdec_amd/build/conv_mfma_t2.o flowdec_amd/build/conv_wino_t2.o flowdec_amd/build/conv_wino4_$NAME.o $OTHERS
echo built flowdec_amd/variants/libflowdec_$NAME.so
