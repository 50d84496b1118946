#!/bin/bash
# side-stream branches (time embedding, pyramid-head chain): parity + bench with and without (FLOWDEC_SIDE_STREAM=0)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_model.py tests/test_hip_configs.py tests/test_hip_baselines.py tests/test_cli.py -m gpu -q -x < /dev/null > $O/pytest_side.log 2>&1; echo "pytest rc=$?" >> $O/pytest_side.log
tail -4 $O/pytest_side.log
Q="--no-cpu-baseline --no-roofline"
for S in 1 0; do
  FLOWDEC_SIDE_STREAM=$S python bench.py $Q --batch 1 --seconds 1 --steps 20 --warmup 3 --conv-algo latency 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('side=$S B=1x1s latency', round(r['value'],2), 'x', round(r['ms_per_step'],2), 'ms')"
  FLOWDEC_SIDE_STREAM=$S python bench.py $Q --batch 1 --seconds 1 --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('side=$S B=1x1s auto', round(r['value'],2), 'x', round(r['ms_per_step'],2), 'ms')"
  FLOWDEC_SIDE_STREAM=$S python bench.py $Q --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('side=$S cfg2', round(r['value'],2), 'x', round(r['ms_per_step'],2), 'ms')"
done | tee $O/bench_side.txt
