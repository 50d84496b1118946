#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
scripts/bin/probe_lane_ops > $O/probe_lane_ops.txt 2>&1; cat $O/probe_lane_ops.txt
for v in t2 renostore; do echo "== $v"; FLOWDEC_HIP_LIB=$GRAFT_REPO_ROOT/flowdec_amd/variants/libflowdec_$v.so timeout 200 python scripts/re_timing2.py 2>&1 | grep -v amdgpu; done | tee $O/re_timing2.txt
