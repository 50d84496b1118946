#!/bin/bash
# FD_LOW_LATENCY: parity of the tile widths / the latency schedule, then B = 1 x 1 s bench per algorithm
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_configs.py -m gpu -q -x -k "tile_widths or winograd_parity" < /dev/null > $O/pytest_latency.log 2>&1; echo "pytest rc=$?" >> $O/pytest_latency.log
tail -5 $O/pytest_latency.log
Q="--no-cpu-baseline --no-roofline"
for A in auto latency winograd; do python bench.py $Q --batch 1 --seconds 1 --steps 20 --warmup 3 --conv-algo $A 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('B=1x1s', '$A', round(r['value'],2), 'x', round(r['ms_per_step'],2), 'ms')"; done | tee $O/bench_latency.txt
for A in auto latency; do python bench.py $Q --batch 1 --seconds 2 --steps 10 --warmup 3 --conv-algo $A 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('B=1x2s', '$A', round(r['value'],2), 'x', round(r['ms_per_step'],2), 'ms')"; done | tee -a $O/bench_latency.txt
