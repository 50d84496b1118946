#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_configs.py tests/test_hip_ops.py -m gpu -q -x -k "register or tile or conv2d" < /dev/null 2>&1 | tail -3
for v in t2 t2stagger; do echo "== $v"; FLOWDEC_HIP_LIB=$GRAFT_REPO_ROOT/flowdec_amd/variants/libflowdec_$v.so timeout 200 python scripts/re_timing2.py 2>&1 | grep -v "amdgpu\|per-workgroup"; done
timeout 900 python scripts/ab_conv_tile.py flowdec_amd/variants/libflowdec_re.so flowdec_amd/variants/libflowdec_restagger.so 2>&1 | tee $O/ab_tile.txt
