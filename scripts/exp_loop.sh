#!/bin/bash
# Ablation builds of the conv kernel ("timing only, WRONG results" blocks FD_EXP_*).  The blocks do not live in the shipping
# source: this script applies scripts/conv_mfma_ablation.patch to a COPY of conv_mfma.hip and builds variant libraries from it
# (scripts/build_variant.sh -> flowdec_amd/variants/libflowdec_NAME.so, selected with FLOWDEC_HIP_LIB); the product library is
# never rebuilt with these flags.  Run the builds locally (hipcc cross-compiles), the timing on the GPU box.
set -e
cd "$(dirname "$0")/.."
mkdir -p flowdec_amd/build/ablation
cp flowdec_amd/csrc/*.h flowdec_amd/csrc/conv_mfma.hip flowdec_amd/build/ablation/
(cd flowdec_amd/build/ablation && patch -p3 < ../../../scripts/conv_mfma_ablation.patch)
i=0
for v in "-DFD_EXP_NOEPI" "-DFD_EXP_NOSTORE" "-DFD_EXP_NOHALO -DFD_EXP_NOEPI" "-DFD_EXP_NOBARRIER -DFD_EXP_NOEPI"; do
  i=$((i + 1))
  bash scripts/build_variant.sh abl$i flowdec_amd/build/ablation/conv_mfma.hip -Iflowdec_amd/build/ablation $v
  echo "abl$i = $v"
done
echo "then on the GPU box: python scripts/ab_conv.py base=flowdec_amd/libflowdec_hip.so abl1=flowdec_amd/variants/libflowdec_abl1.so ..."
