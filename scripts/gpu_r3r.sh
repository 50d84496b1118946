#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_configs.py -m gpu -q -k "latency or tile" < /dev/null 2>&1 | tail -3
Q="--no-cpu-baseline --no-e2e"
for i in 1 2; do python bench.py $Q --batch 1 --seconds 1 --steps 30 --warmup 5 --conv-algo latency > $O/bench_b1_latency.json 2>/dev/null; python -c "import json; r=json.load(open('$O/bench_b1_latency.json')); print('b1 latency', round(r['value'],2), round(r['ms_per_step'],2))"; done
python bench.py $Q --batch 1 --seconds 2 --steps 20 --warmup 5 --conv-algo latency > $O/bench_b1_2s_latency.json 2>/dev/null; python -c "import json; r=json.load(open('$O/bench_b1_2s_latency.json')); print('b1 2s latency', round(r['value'],2), round(r['ms_per_step'],2))"
python bench.py $Q --batch 1 --seconds 2 --steps 20 --warmup 5 > $O/bench_b1_2s_auto.json 2>/dev/null; python -c "import json; r=json.load(open('$O/bench_b1_2s_auto.json')); print('b1 2s auto', round(r['value'],2), round(r['ms_per_step'],2))"
