#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x < /dev/null > $O/pytest_small.log 2>&1; echo "pytest rc=$?" >> $O/pytest_small.log
tail -5 $O/pytest_small.log
Q="--no-cpu-baseline --no-roofline"
for A in auto latency; do python bench.py $Q --batch 1 --seconds 1 --steps 20 --warmup 3 --conv-algo $A 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('B=1x1s', '$A', round(r['value'],2), 'x', round(r['ms_per_step'],2), 'ms')"; done | tee $O/bench_small.txt
python bench.py $Q --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('cfg2 auto', round(r['value'],2), 'x', round(r['ms_per_step'],2), 'ms')" | tee -a $O/bench_small.txt
rm -rf $O/prof_b1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b1 -- python $GRAFT_REPO_ROOT/bench.py --batch 1 --seconds 1 --steps 10 --warmup 3 --conv-algo latency --no-cpu-baseline --no-roofline < /dev/null > $O/prof_b1.log 2>&1); echo "prof rc=$?"
python scripts/step_breakdown.py $O/prof_b1/*/*_kernel_trace.csv 60 > $O/b1_latency_breakdown.txt; head -30 $O/b1_latency_breakdown.txt
