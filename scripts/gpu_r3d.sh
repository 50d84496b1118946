#!/bin/bash
# round 3, call D: register-epilogue kernel after the lane-swap fix: probe, parity, A/B, counters (cycles vs clock), benches
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; rm -f $O/parity_report.txt
scripts/bin/probe_lane_ops > $O/probe_lane_ops.txt 2>&1; tail -12 $O/probe_lane_ops.txt
timeout 1200 python -m pytest tests -m gpu -q -x --deselect tests/test_hip_configs.py::test_cfg5_fp32_4s_32step_and_adaptive < /dev/null > $O/pytest_re.log 2>&1; echo "pytest rc=$?" >> $O/pytest_re.log
tail -15 $O/pytest_re.log
timeout 600 python scripts/ab_conv.py nore=flowdec_amd/variants/libflowdec_nore.so re=flowdec_amd/variants/libflowdec_re.so --rounds 3 > $O/ab_re.txt 2>&1; cat $O/ab_re.txt
SHAPE=1 bash scripts/pmc_ab.sh nore re 2>&1 | grep -v "^time" | tee $O/pmc_ab_re_plain.txt
SHAPE=2 bash scripts/pmc_ab.sh nore re 2>&1 | grep -v "^time" | tee $O/pmc_ab_re_cat.txt
Q="--no-cpu-baseline --no-roofline --no-e2e"
for V in re nore; do
  export FLOWDEC_HIP_LIB=$GRAFT_REPO_ROOT/flowdec_amd/variants/libflowdec_$V.so
  python bench.py $Q --steps 8 --warmup 3 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('$V cfg2', round(r['value'],2), 'x', round(r['ms_per_step'],2), 'ms')"
  python bench.py $Q --batch 1 --seconds 1 --steps 30 --warmup 5 --conv-algo latency 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('$V B=1x1s latency', round(r['value'],2), 'x', round(r['ms_per_step'],2), 'ms')"
  python bench.py $Q --batch 1 --seconds 1 --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('$V B=1x1s auto', round(r['value'],2), 'x', round(r['ms_per_step'],2), 'ms')"
  python bench.py $Q --steps 8 --warmup 3 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('$V cfg2 again', round(r['value'],2), 'x', round(r['ms_per_step'],2), 'ms')"
done | tee $O/bench_re_vs_nore.txt
