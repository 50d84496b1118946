#!/bin/bash
# round 6, call J: the 2-D float32 Winograd kernel (conv_wino44f.hip): operator parity, fp32 model tests, fp32 bench lines
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; rm -f $O/parity_report.txt
timeout 900 python -m pytest tests/test_hip_configs.py -m gpu -q -k "winograd4_f32 or fp32_auto" < /dev/null > $O/r6j_w44.log 2>&1; echo "w44 rc=$?" >> $O/r6j_w44.log; tail -30 $O/r6j_w44.log | cut -c1-220
grep "winograd44" $O/parity_report.txt | cut -c1-110
timeout 1200 python -m pytest tests -m gpu -q -k "fp32" < /dev/null > $O/r6j_fp32.log 2>&1; echo "fp32 rc=$?" >> $O/r6j_fp32.log; tail -6 $O/r6j_fp32.log | cut -c1-200
grep "fp32" $O/parity_report.txt | grep -v "conv2d_" | cut -c1-110 | sort -u | head -20
for algo in auto direct; do
  timeout 600 python bench.py --precision fp32 --conv-algo $algo --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r6j_bench_fp32_$algo.json
  python -c "import json; j=json.load(open('$O/r6j_bench_fp32_$algo.json')); print('fp32 cfg2-shape $algo', round(j['value'],2), 'x', round(j['ms_per_step'],1), 'ms frac', round(j['roofline']['frac'],3), 'exec', round(j['roofline']['executed_frac_of_peak'],3))"
done
timeout 900 python bench.py --config cfg5 --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r6j_bench_cfg5.json
python -c "import json; j=json.load(open('$O/r6j_bench_cfg5.json')); print('cfg5 shard', round(j['value'],2), 'x', round(j['ms_per_step'],1), 'ms')"
