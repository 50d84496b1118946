#!/bin/bash
# round 5, call E: resident workgroups that walk the tiles (W4_PERSIST = 256: one per CU; 512: two waves of dispatch) against the plain grid
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
for v in hip persist; do
  L=$GRAFT_REPO_ROOT/flowdec_amd/variants/libflowdec_$v.so; [ $v = hip ] && L=$GRAFT_REPO_ROOT/flowdec_amd/libflowdec_hip.so
  echo "== $v"; FLOWDEC_HIP_LIB=$L timeout 400 python scripts/wino4_check.py 2>&1 | grep -v amdgpu.ids | tail -22
done > $O/r5e_wino4_check.txt 2>&1; grep "^==\|^time\|FAIL\|parity" $O/r5e_wino4_check.txt | cut -c1-150
bash scripts/ab_bench_libs.sh 3 hip persist persist512 2>&1 | tee $O/r5e_ab_persist.txt
