#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
run() { echo "=== variant: $1"; FLOWDEC_EXTRA_FLAGS="$1" python flowdec_amd/build.py --force > /dev/null 2>&1 < /dev/null || echo BUILD FAILED; timeout 200 python scripts/conv_bench.py --B 8 --iters 20 2>&1 | grep -v amdgpu | head -12; }
timeout 300 python -m pytest tests/test_hip_ops.py -q -x -k "conv2d or resblock" 2>&1 | tail -3
run "" 2>&1 | tee $O/direct_v0.log
run "-DFD_PIN" 2>&1 | tee $O/direct_pin.log
run "-DFD_PIN -DFD_SETPRIO" 2>&1 | tee $O/direct_pin_prio.log
python flowdec_amd/build.py --force > /dev/null 2>&1
