#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; rm -f $O/parity_report.txt
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_configs.py -m gpu -q -x -k "conv2d or resblock or tile or register or winograd_parity or cfg2 or threads" < /dev/null > $O/pytest_re.log 2>&1; echo "pytest rc=$?" >> $O/pytest_re.log
tail -6 $O/pytest_re.log
echo "== staged"; FLOWDEC_HIP_LIB=$GRAFT_REPO_ROOT/flowdec_amd/variants/libflowdec_t2nore.so timeout 200 python scripts/conv_timing2.py 2>&1 | grep -v amdgpu
echo "== RE"; FLOWDEC_HIP_LIB=$GRAFT_REPO_ROOT/flowdec_amd/variants/libflowdec_t2.so timeout 200 python scripts/re_timing2.py 2>&1 | grep -v "amdgpu\|per-workgroup"
timeout 600 python scripts/ab_conv.py nore=flowdec_amd/variants/libflowdec_nore.so re=flowdec_amd/variants/libflowdec_re.so --rounds 3 > $O/ab_re2.txt 2>&1; cat $O/ab_re2.txt
SHAPE=1 bash scripts/pmc_ab.sh nore re 2>&1 | grep -v "^time" | tee $O/pmc_ab_re2_plain.txt
timeout 600 python -m pytest tests/test_hip_ndac.py -m gpu -q < /dev/null > $O/pytest_ndac.log 2>&1; echo "pytest rc=$?" >> $O/pytest_ndac.log; tail -30 $O/pytest_ndac.log
