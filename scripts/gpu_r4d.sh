#!/bin/bash
# round 4, call D: SQ / LDS counters of F(4,3) kernel variants (timing-only ablations) on one shape
cd $GRAFT_REPO_ROOT
for v in ${VARIANTS:-none noprod notrans}; do
  echo "=== variant $v"
  FLOWDEC_HIP_LIB=$GRAFT_REPO_ROOT/flowdec_amd/variants/libflowdec_$v.so SHAPE=${SHAPE:-1} bash scripts/pmc_wino4.sh 2>&1 | grep -A30 "conv_wino4" | grep "LDS_BANK\|LDS_IDX\|INSTS_LDS\|cycles/dispatch\|wave time\|FIFO"
done
