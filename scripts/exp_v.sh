#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in "-DFD_NO_SGB" "-DFD_SETPRIO" "-DFD_HLAG=1"; do
  echo "=== build [$v]"
  FLOWDEC_EXTRA_FLAGS="$v" python flowdec_amd/build.py --force > /dev/null 2>&1 < /dev/null || echo BUILD FAILED
  for cv in 0 3; do echo "-- conv_variant=$cv"; for o in 0 1 2; do timeout 120 python scripts/conv_bench.py --B 8 --iters 40 --only $o --variant $cv < /dev/null 2>&1 | grep TFLOP; done; done
done
python flowdec_amd/build.py --force > /dev/null 2>&1 < /dev/null
