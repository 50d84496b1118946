#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in 81 161 41; do echo "VEC8 N*10+BX=$v"; FD_EXP_UP=$v timeout 300 python scripts/fir_bench.py 2>&1 | grep "dir +1"; done
for v in 81 161 41 82 42; do echo "VEC4 N*10+BX=$v"; FD_EXP_UP4=1 FD_EXP_UP=$v timeout 300 python scripts/fir_bench.py 2>&1 | grep "dir +1"; done
