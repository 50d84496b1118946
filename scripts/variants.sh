#!/bin/bash
# build conv variants on the GPU box and run the micro-benchmark for each
cd $GRAFT_REPO_ROOT
for v in "" "-DFD_SETPRIO" "-DFD_NO_SGB" "-DFD_HLAG=1" "-DFD_HLAG=3"; do
  echo "=== variant [$v]"
  FLOWDEC_EXTRA_FLAGS="$v" python flowdec_amd/build.py --force > /dev/null 2>&1 < /dev/null || echo BUILD FAILED
  for it in 1 2; do timeout 120 python scripts/conv_bench.py --iters 10 --only 1 < /dev/null 2>&1 | grep TFLOP; timeout 120 python scripts/conv_bench.py --iters 10 --only 2 < /dev/null 2>&1 | grep TFLOP; done
done
python flowdec_amd/build.py --force > /dev/null 2>&1 < /dev/null
