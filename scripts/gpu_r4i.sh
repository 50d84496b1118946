#!/bin/bash
# round 4, call I: one clip (B = 1 x 1 s and 1 x 2 s) per conv schedule after the F(4,3) kernel joined `auto` and `latency`
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
for algo in auto latency; do for sec in 1 2; do
  timeout 300 python bench.py --batch 1 --seconds $sec --conv-algo $algo --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r4i_b1_${sec}s_$algo.json
  python -c "import json,sys; j=json.load(open('$O/r4i_b1_${sec}s_$algo.json')); print('$algo', $sec, 's:', round(j['value'],2), 'x', round(j['ms_per_step'],3), 'ms; conv', round(j['roofline']['conv_ms_per_step'],2), 'ms, frac', round(j['roofline']['frac'],3))"
done; done
timeout 600 python -m pytest tests/test_hip_configs.py -q -x -m gpu -k "latency" 2>&1 | tail -3
