#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in "" "-DFD_EXP_NOSTORE" "-DFD_EXP_NOEPI"; do
  echo "=== variant [$v]"
  FLOWDEC_EXTRA_FLAGS="$v" python flowdec_amd/build.py --force > /dev/null 2>&1 < /dev/null || echo BUILD FAILED
  for B in 2 8; do for cv in 0 1; do echo "-- B=$B conv_variant=$cv"; for o in 0 1 2; do timeout 120 python scripts/conv_bench.py --B $B --iters 10 --only $o --variant $cv < /dev/null 2>&1 | grep TFLOP; done; done; done
done
python flowdec_amd/build.py --force > /dev/null 2>&1 < /dev/null
