#!/bin/bash
# round 5, call B: prologue line-prefetch variants of conv_wino4 (scripts/build_w4prod.sh pf / pf1): parity + per-launch A/B on the layer
# shapes, then interleaved cfg 2 bench lines in the same call
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
for v in hip pf pf1; do
  L=$GRAFT_REPO_ROOT/flowdec_amd/variants/libflowdec_$v.so; [ $v = hip ] && L=$GRAFT_REPO_ROOT/flowdec_amd/libflowdec_hip.so
  echo "== $v"; FLOWDEC_HIP_LIB=$L timeout 400 python scripts/wino4_check.py 2>&1 | grep -v amdgpu.ids | tail -22
done > $O/r5b_wino4_check.txt 2>&1; cat $O/r5b_wino4_check.txt
bash scripts/ab_bench_libs.sh 3 hip pf pf1 2>&1 | tee $O/r5b_ab_pf.txt
