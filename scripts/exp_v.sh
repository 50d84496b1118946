#!/bin/bash
cd $GRAFT_REPO_ROOT
for cv in 0 2; do echo "-- conv_variant=$cv"; for o in 0 1 2 5; do timeout 120 python scripts/conv_bench.py --B 8 --iters 10 --only $o --variant $cv < /dev/null 2>&1 | grep TFLOP; done; done
