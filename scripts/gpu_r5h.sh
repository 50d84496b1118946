#!/bin/bash
# round 5, call H: the new reference goldens at cfg 4's / cfg 5's own sizes (G24, G25), the scale-fold variant of the F(4,3) epilogue
# (-DW4_SCALEFOLD: bias and inverse weight scale pre-multiplied by `scale`, one fma instead of fma + mul per output pair) as parity + A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; rm -f $O/parity_report.txt
timeout 600 python -m pytest tests -m gpu -q -x -k "cfg4_image or cfg5_clip or cfg3_image or cfg2_image" < /dev/null 2>&1 | tail -4
V=$GRAFT_REPO_ROOT/flowdec_amd/variants
FLOWDEC_HIP_LIB=$V/libflowdec_sfold.so timeout 600 python -m pytest tests -m gpu -q -x -k "winograd4 or wino4" < /dev/null 2>&1 | tail -3
for v in hip sfold; do
  L=$V/libflowdec_$v.so; [ $v = hip ] && L=$GRAFT_REPO_ROOT/flowdec_amd/libflowdec_hip.so
  echo "== $v"; FLOWDEC_HIP_LIB=$L timeout 400 python scripts/wino4_check.py --no-parity 2>&1 | grep "^time"
done 2>&1 | cut -c1-140 | tee $O/r5h_wino4_sfold.txt
bash scripts/ab_bench_libs.sh 3 hip sfold 2>&1 | tee $O/r5h_ab_sfold.txt
cp $O/parity_report.txt $O/r5h_parity_new_goldens.txt 2>/dev/null
