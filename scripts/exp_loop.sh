#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in "" "-DFD_EXP_NOPROWAIT" "-DFD_EXP_NOPROWAIT -DFD_EXP_NOEPI"; do
  echo "=== variant [$v]"
  FLOWDEC_EXTRA_FLAGS="$v" python flowdec_amd/build.py --force > /dev/null 2>&1 < /dev/null || echo BUILD FAILED
  for o in 0 1 2 5; do timeout 120 python scripts/conv_bench.py --B 8 --iters 10 --only $o < /dev/null 2>&1 | grep TFLOP; done
done
python flowdec_amd/build.py --force > /dev/null 2>&1 < /dev/null
