#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
Q="--no-cpu-baseline --no-roofline --steps 5 --warmup 2"
for A in direct auto winograd_lowres; do python bench.py $Q --conv-algo $A 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('cfg2 B=8x2s', '$A', round(r['value'],2), 'x', round(r['ms_per_step'],2), 'ms')"; done
for A in direct auto winograd; do python bench.py $Q --batch 1 --seconds 1 --steps 20 --warmup 3 --conv-algo $A 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('B=1x1s', '$A', round(r['value'],2), 'x', round(r['ms_per_step'],2), 'ms')"; done
for A in direct auto; do python bench.py $Q --batch 2 --seconds 2 --steps 10 --warmup 3 --conv-algo $A 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('B=2x2s', '$A', round(r['value'],2), 'x', round(r['ms_per_step'],2), 'ms')"; done
