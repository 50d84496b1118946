#!/bin/bash
# round 6, call I: power-aware A/B of non-temporal stores in the elementwise kernels (FIR resampling, Combine, ...: fd_store_vec)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
FLOWDEC_HIP_LIB=$GRAFT_REPO_ROOT/flowdec_amd/variants/libflowdec_ewnt.so timeout 600 python -m pytest tests/test_hip_ops.py -m gpu -q -k "fir or combine or resblock" < /dev/null 2>&1 | tail -3
bash scripts/ab_bench_libs.sh 3 hip ewnt 2>&1 | tee $O/r6i_ab_ewnt.txt
