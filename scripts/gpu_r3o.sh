#!/bin/bash
# FIR kernels: parity tests + micro-benchmark
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_ops.py -m gpu -q -k "fir or resblock or upfirdn" < /dev/null 2>&1 | tail -3
timeout 300 python scripts/fir_bench.py 2>&1 | grep dir
