#!/bin/bash
# round 4, call C: model-level parity with the F(4,3) kernel in the default schedule + the cfg 2 bench line
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_model.py tests/test_hip_ops.py -x -q -m gpu 2>&1 | tail -5
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/r4c_bench.log
