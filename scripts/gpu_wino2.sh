#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 300 python scripts/wino_check.py --B 8 --iters 10 --rounds 3 > $O/wino_check.log 2>&1; echo "wino rc=$?"
cat $O/wino_check.log | grep -v amdgpu.ids | tail -40
FLOWDEC_EXTRA_FLAGS="-DFD_TIMING2" python flowdec_amd/build.py --force > /dev/null 2>&1 < /dev/null || echo BUILD FAILED
timeout 200 python scripts/wino_timing2.py < /dev/null 2>&1 | grep -v amdgpu | tee $O/wino_timing2.log
timeout 200 python scripts/conv_timing2.py < /dev/null 2>&1 | grep -v amdgpu | tee $O/conv_timing2.log
