#!/bin/bash
# round 4, call B: what bounds the F(4,3) K loop?  timing-only variants (wrong results)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
for v in ${VARIANTS:-t2 noconv notrans nohalo noprod}; do
  echo "== variant $v"
  FLOWDEC_HIP_LIB=$GRAFT_REPO_ROOT/flowdec_amd/variants/libflowdec_$v.so timeout 200 python scripts/wino4_timing2.py < /dev/null 2>&1 | grep wino4
done | tee $O/r4b_variants.log
