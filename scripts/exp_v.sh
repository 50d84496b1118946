#!/bin/bash
cd $GRAFT_REPO_ROOT
for sg in 0 8 16 32 64; do echo "-- stagger=$sg"; for o in 0 1 2; do timeout 120 python scripts/conv_bench.py --B 8 --iters 40 --only $o --stagger $sg < /dev/null 2>&1 | grep TFLOP; done; done
