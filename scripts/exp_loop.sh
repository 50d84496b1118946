#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in "-DFD_EXP_NOEPI" "-DFD_EXP_NOEPI -DFD_EXP_NOBARRIER" "-DFD_EXP_NOEPI -DFD_EXP_NODMA" "-DFD_EXP_NOEPI -DFD_EXP_NOHALO" "-DFD_EXP_NOEPI -DFD_EXP_NOHALO -DFD_EXP_NODMA" "-DFD_EXP_NOEPI -DFD_EXP_NOHALO -DFD_EXP_NODMA -DFD_EXP_NOBARRIER"; do
  echo "=== variant [$v]"
  FLOWDEC_EXTRA_FLAGS="$v" python flowdec_amd/build.py --force > /dev/null 2>&1 < /dev/null || echo BUILD FAILED
  for o in 1 2; do timeout 120 python scripts/conv_bench.py --B 8 --iters 10 --only $o < /dev/null 2>&1 | grep TFLOP; done
done
python flowdec_amd/build.py --force > /dev/null 2>&1 < /dev/null
