#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "conv" < /dev/null 2>&1 | tail -2
for o in 11; do timeout 120 python scripts/conv_bench.py --B 8 --iters 20 --only $o < /dev/null 2>&1 | grep TFLOP; done
timeout 300 python bench.py --no-cpu-baseline < /dev/null 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
