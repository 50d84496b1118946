#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
for i in 1 2; do
echo "== staged epilogue (FD_NO_RE), one workgroup per tile"
FLOWDEC_HIP_LIB=$GRAFT_REPO_ROOT/flowdec_amd/variants/libflowdec_t2nore.so timeout 200 python scripts/conv_timing2.py 2>&1 | grep -v amdgpu
echo "== register epilogue, continuous tiles"
FLOWDEC_HIP_LIB=$GRAFT_REPO_ROOT/flowdec_amd/variants/libflowdec_t2.so timeout 200 python scripts/re_timing2.py 2>&1 | grep -v "amdgpu\|per-workgroup"
done | tee $O/re_vs_staged_timing2.txt
