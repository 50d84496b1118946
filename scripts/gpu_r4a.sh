#!/bin/bash
# round 4, call A: F(4,3) kernel parity + per-launch timing vs the direct kernel, phase split of both
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 400 python scripts/wino4_check.py --B 8 --iters 10 --rounds 2 > $O/r4a_wino4_check.log 2>&1; echo "wino4_check rc=$?"
grep -v amdgpu.ids $O/r4a_wino4_check.log | tail -60
FLOWDEC_HIP_LIB=$GRAFT_REPO_ROOT/flowdec_amd/variants/libflowdec_t2.so timeout 200 python scripts/wino4_timing2.py < /dev/null 2>&1 | grep -v amdgpu | tee $O/r4a_wino4_timing2.log
