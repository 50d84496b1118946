#!/bin/bash
# round 5, call F: resident workgroups + the next tile's first halo requested under the epilogue (W4_PERSIST builds of conv_wino4.hip):
# parity with 8 resident workgroups (every workgroup walks many tiles, also in the small test cases), race hunt, per-launch timing and the
# cfg 2 A/B with 256 resident workgroups
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
V=$GRAFT_REPO_ROOT/flowdec_amd/variants
FLOWDEC_HIP_LIB=$V/libflowdec_persist8.so timeout 900 python -m pytest tests -m gpu -q -x -k "winograd4 or wino4 or weight_range or full_width or cfg2_image or cfg1_exact or bf16_error" < /dev/null > $O/r5f_pytest_persist8.log 2>&1; tail -4 $O/r5f_pytest_persist8.log
FLOWDEC_HIP_LIB=$V/libflowdec_persist8.so timeout 600 python scripts/wino4_race.py 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/r5f_race_persist8.txt
FLOWDEC_HIP_LIB=$V/libflowdec_persist.so timeout 600 python scripts/wino4_race.py 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/r5f_race_persist.txt
for v in hip persist; do
  L=$V/libflowdec_$v.so; [ $v = hip ] && L=$GRAFT_REPO_ROOT/flowdec_amd/libflowdec_hip.so
  echo "== $v"; FLOWDEC_HIP_LIB=$L timeout 400 python scripts/wino4_check.py 2>&1 | grep -v amdgpu.ids | tail -22
done > $O/r5f_wino4_check.txt 2>&1; grep "^==\|^time\|FAIL" $O/r5f_wino4_check.txt | cut -c1-150
bash scripts/ab_bench_libs.sh 3 hip persist 2>&1 | tee $O/r5f_ab_persist.txt
