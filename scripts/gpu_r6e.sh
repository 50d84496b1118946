#!/bin/bash
# round 6, call E: delta probes of the float32 F(4,3) kernel + its tests
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 300 python scripts/wino4f_debug.py > $O/r6e_w4f_debug.txt 2>&1; head -120 $O/r6e_w4f_debug.txt
timeout 900 python -m pytest tests/test_hip_configs.py -m gpu -q -k "winograd4_f32 or fp32_auto" < /dev/null > $O/r6e_w4f.log 2>&1; echo "w4f rc=$?" >> $O/r6e_w4f.log; tail -25 $O/r6e_w4f.log | cut -c1-200
