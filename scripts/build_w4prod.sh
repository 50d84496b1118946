#!/bin/bash
# build_w4prod.sh NAME FLAGS...: a PRODUCT-flag variant library whose conv_wino4.hip is compiled with extra -D flags (experiment switches
# exist after `patch -p0 < scripts/conv_wino4_r5_experiments.patch`); everything else = the objects of the current build.
# -> flowdec_amd/variants/libflowdec_NAME.so (travels with gpurun; select with FLOWDEC_HIP_LIB, A/B with scripts/ab_bench_libs.sh)
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p flowdec_amd/variants
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Iinclude -Iflowdec_amd/csrc -Wno-unused-result"
/opt/rocm/bin/hipcc $F "$@" -c flowdec_amd/csrc/conv_wino4.hip -o flowdec_amd/build/conv_wino4_$NAME.o
OTHERS=$(ls flowdec_amd/build/{api,calib,conv_mfma,conv_wino,conv_head,elementwise,stft,model,ndac,ndac_mfma}.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o flowdec_amd/variants/libflowdec_$NAME.so flowdec_amd/build/conv_wino4_$NAME.o $OTHERS
echo built flowdec_amd/variants/libflowdec_$NAME.so
