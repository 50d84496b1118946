#!/bin/bash
# Extra bench lines for the record (not the driver's default): fp32 mode, cfg3 (25s preset, B=32, midpoint NFE 6), B=1 / B=2 with the
# conv algorithm choices, cfg5-like.
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
run() { name=$1; shift; timeout 400 python bench.py "$@" --no-cpu-baseline < /dev/null > $O/bench_$name.json 2> $O/bench_$name.err; python -c "
import json,sys
d=json.load(open('$O/bench_$name.json')); r=d.get('roofline',{})
print('$name', round(d['value'],2), 'x', round(d['ms_per_step'],1), 'ms', round(r.get('achieved',0),1), 'TF', d['config']['workload'][:70])" ; }
run fp32 --precision fp32 --steps 2 --warmup 2
run bf16x3 --precision bf16x3 --steps 3 --warmup 2
run mixed --precision mixed --steps 3 --warmup 2
run cfg3 --preset flowdec_25s --batch 32 --solver midpoint --N 3 --steps 2 --warmup 2
run cfg2_direct --conv-algo direct
run b1_direct --batch 1 --seconds 1 --steps 20 --warmup 3 --conv-algo direct
run b1_auto --batch 1 --seconds 1 --steps 20 --warmup 3
run b1_winograd --batch 1 --seconds 1 --steps 20 --warmup 3 --conv-algo winograd
run b1_latency --batch 1 --seconds 1 --steps 20 --warmup 3 --conv-algo latency
run b2_direct --batch 2 --seconds 2 --steps 10 --warmup 3 --conv-algo direct
run b2_winograd --batch 2 --seconds 2 --steps 10 --warmup 3 --conv-algo winograd
run cfg5like --precision fp32 --batch 8 --seconds 4 --N 32 --steps 1 --warmup 2 --no-roofline
