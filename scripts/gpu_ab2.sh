#!/bin/bash
cd $GRAFT_REPO_ROOT
V=flowdec_amd/variants
FLOWDEC_HIP_LIB=$V/libflowdec_asm.so timeout 300 python -m pytest tests/test_hip_ops.py -q -x -k "conv2d" 2>&1 | tail -2
python scripts/ab_conv.py old=$V/libflowdec_old.so asm=$V/libflowdec_asm.so asmprio=$V/libflowdec_asmprio.so --rounds 2 2>&1 | grep -v amdgpu | tee gpurun_out/ab2.log
