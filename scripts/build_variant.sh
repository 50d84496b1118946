#!/bin/bash
# build_variant.sh NAME SRC FLAGS... : compile one conv source variant locally (hipcc cross-compiles) and link it against the
# objects of the current build -> flowdec_amd/variants/libflowdec_NAME.so (travels with gpurun; select with FLOWDEC_HIP_LIB).
set -e
cd "$(dirname "$0")/.."
NAME=$1; SRC=$2; shift 2
mkdir -p flowdec_amd/variants flowdec_amd/build
OBJ=flowdec_amd/build/conv_mfma_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Iinclude -Iflowdec_amd/csrc -Wno-unused-result "$@" -c $SRC -o $OBJ
OTHERS=$(ls flowdec_amd/build/{api,conv_wino,conv_wino4,conv_head,elementwise,stft,model,ndac,ndac_mfma}.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o flowdec_amd/variants/libflowdec_$NAME.so $OBJ $OTHERS
echo built flowdec_amd/variants/libflowdec_$NAME.so
