#!/bin/bash
# FIR kernels: parity tests + micro-benchmark (+ the f32-storage modes after a FIR change)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_ops.py -m gpu -q -k "fir or resblock or upfirdn" < /dev/null 2>&1 | tail -3
timeout 300 python scripts/fir_bench.py 2>&1 | grep dir
Q="--no-cpu-baseline --no-e2e"
python bench.py $Q --precision mixed --steps 3 --warmup 1 > $O/bench_mixed.json 2>/dev/null; python -c "import json; r=json.load(open('$O/bench_mixed.json')); print('mixed', round(r['value'],2), round(r['ms_per_step'],1), round(r['roofline_hbm']['ms_per_step'],2))"
python bench.py $Q --precision bf16x3 --steps 3 --warmup 1 > $O/bench_bf16x3.json 2>/dev/null; python -c "import json; r=json.load(open('$O/bench_bf16x3.json')); print('bf16x3', round(r['value'],2), round(r['ms_per_step'],1), round(r['roofline_hbm']['ms_per_step'],2))"
