#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
echo "VEC8 8x1"; timeout 300 python scripts/fir_bench.py 2>&1 | grep "dir +1"
for v in 3 4; do echo "VEC8 exp=$v"; FD_EXP_UP=$v timeout 300 python scripts/fir_bench.py 2>&1 | grep "dir +1"; done
for v in 1 2 3 4; do echo "VEC4 exp=$v (1 = 8x1, 2 = 8x2, 3 = 4x1, 4 = 16x1)"; FD_EXP_UP4=1 FD_EXP_UP=$v timeout 300 python scripts/fir_bench.py 2>&1 | grep "dir +1"; done
