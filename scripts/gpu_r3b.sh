#!/bin/bash
# round 3, call B: register-epilogue / continuous-tile conv kernel (RE): parity, same-box A/B against the staged-epilogue build
# (-DFD_NO_RE variant), bench line.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O; rm -f $O/parity_report.txt
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_configs.py -m gpu -q -x -k "conv2d or resblock or tile or register or winograd_parity or cfg2 or threads" < /dev/null > $O/pytest_re.log 2>&1; echo "pytest rc=$?" >> $O/pytest_re.log
tail -25 $O/pytest_re.log
timeout 600 python scripts/ab_conv.py nore=flowdec_amd/variants/libflowdec_nore.so re=flowdec_amd/libflowdec_hip.so --rounds 2 > $O/ab_re.txt 2>&1; cat $O/ab_re.txt
timeout 300 python bench.py --no-cpu-baseline < /dev/null > $O/bench_re.json 2> $O/bench_re.err; python -c "
import json; r=json.load(open('$O/bench_re.json')); print('RE  value', round(r['value'],2), 'ms', round(r['ms_per_step'],2), 'conv ms', round(r['roofline']['conv_ms_per_step'],2), 'frac', round(r['roofline']['frac'],4))"
FLOWDEC_HIP_LIB=$GRAFT_REPO_ROOT/flowdec_amd/variants/libflowdec_nore.so timeout 300 python bench.py --no-cpu-baseline < /dev/null > $O/bench_nore.json 2> $O/bench_nore.err; python -c "
import json; r=json.load(open('$O/bench_nore.json')); print('noRE value', round(r['value'],2), 'ms', round(r['ms_per_step'],2), 'conv ms', round(r['roofline']['conv_ms_per_step'],2), 'frac', round(r['roofline']['frac'],4))"
