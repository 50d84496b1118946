#!/bin/bash
# round 5, call C: do two concurrent half-batches fill the chip better than one batch?  Two ranks SHARING the GPU (gloo gather on the
# host, reported apart) with 4 and with 8 clips each, against the single-process line, all in one call
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
P="import json,sys; j=json.loads(sys.stdin.readline()); print(sys.argv[1], 'value', round(j['value'],1), 'ms/step', round(j['ms_per_step'],2), 'per-rank', [round(x,1) for x in j['per_rank_ms_per_step']], 'gather', round(j['allgather_ms_per_step'],2))"
for i in 1 2; do
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-e2e 2>/dev/null | grep '^{' | python -c "$P" single_b8
timeout 300 python bench.py --gpus 2 --share-gpu --backend gloo --global-batch 8 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-e2e 2>/dev/null | grep '^{' | python -c "$P" two_ranks_b4+b4
timeout 300 python bench.py --gpus 2 --share-gpu --backend gloo --global-batch 16 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-e2e 2>/dev/null | grep '^{' | python -c "$P" two_ranks_b8+b8
timeout 300 python bench.py --gpus 4 --share-gpu --backend gloo --global-batch 16 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-e2e 2>/dev/null | grep '^{' | python -c "$P" four_ranks_b4x4
done 2>&1 | tee $O/r5c_shared_gpu.txt
