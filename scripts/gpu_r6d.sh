#!/bin/bash
# round 6, call D: the float32 F(4,3) kernel (conv_wino4f.hip): operator parity, the fp32 model tests that now run through it, per-launch
# timing against the direct f32 kernel, and the fp32 bench lines (cfg 2 shape and cfg 5's shard) with `auto` and `direct`.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_configs.py -m gpu -q -x -k "winograd4_f32 or fp32_auto or rejects_unsupported" < /dev/null > $O/r6d_w4f.log 2>&1; echo "w4f rc=$?" >> $O/r6d_w4f.log; tail -15 $O/r6d_w4f.log
timeout 1200 python -m pytest tests -m gpu -q -x -k "fp32" < /dev/null > $O/r6d_fp32.log 2>&1; echo "fp32 rc=$?" >> $O/r6d_fp32.log; tail -8 $O/r6d_fp32.log
for algo in auto direct; do
  timeout 600 python bench.py --precision fp32 --conv-algo $algo --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r6d_bench_fp32_$algo.json
  python -c "import json; j=json.load(open('$O/r6d_bench_fp32_$algo.json')); print('fp32 cfg2-shape $algo', round(j['value'],2), 'x', round(j['ms_per_step'],1), 'ms frac', round(j['roofline']['frac'],3), 'exec', round(j['roofline']['executed_frac_of_peak'],3))"
done
for algo in auto direct; do
  timeout 900 python bench.py --config cfg5 --conv-algo $algo --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r6d_bench_cfg5_$algo.json
  python -c "import json; j=json.load(open('$O/r6d_bench_cfg5_$algo.json')); print('cfg5 shard $algo', round(j['value'],2), 'x', round(j['ms_per_step'],1), 'ms frac', round(j['roofline']['frac'],3))"
done
