#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 120 scripts/bin/mfma_ceiling > $O/mfma_ceiling.log 2>&1; echo "ceiling rc=$?"
timeout 300 python scripts/wino_check.py --B 8 --iters 10 --rounds 3 > $O/wino_check.log 2>&1; echo "wino rc=$?"
timeout 300 python -m pytest tests/test_hip_ops.py -q -x -k "conv2d" > $O/t_conv.log 2>&1; echo "pytest rc=$?"
tail -5 $O/t_conv.log; cat $O/mfma_ceiling.log; cat $O/wino_check.log | tail -40
