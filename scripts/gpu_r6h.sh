#!/bin/bash
# round 6, call H: the split-bf16 (bf16x3) variant of the float32-storage F(4,3) kernel: matrix-instruction rates, operator parity, the
# bf16x3 model tests, bench lines of bf16x3 with `auto` (F(4,3) x3) and `direct`, and the fp32 line again (the f32 path shares the source)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
scripts/bin/mfma_x8_rate 2>&1 | tee $O/r6h_mfma_x8_rate.txt
timeout 900 python -m pytest tests/test_hip_configs.py -m gpu -q -k "winograd4_bf16x3 or winograd4_f32" < /dev/null > $O/r6h_x3.log 2>&1; echo "x3 rc=$?" >> $O/r6h_x3.log; tail -6 $O/r6h_x3.log | cut -c1-200
grep "bf16x3\[" $O/parity_report.txt | cut -c1-120
timeout 1500 python -m pytest tests -m gpu -q -k "bf16x3" < /dev/null > $O/r6h_x3_model.log 2>&1; echo "x3 model rc=$?" >> $O/r6h_x3_model.log; tail -6 $O/r6h_x3_model.log | cut -c1-200
grep "bf16x3" $O/parity_report.txt | grep -v "conv2d_" | cut -c1-120 | head -30
for algo in auto direct; do
  timeout 600 python bench.py --precision bf16x3 --conv-algo $algo --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r6h_bench_bf16x3_$algo.json
  python -c "import json; j=json.load(open('$O/r6h_bench_bf16x3_$algo.json')); print('bf16x3 $algo', round(j['value'],2), 'x', round(j['ms_per_step'],1), 'ms frac', round(j['roofline']['frac'],3), 'exec', round(j['roofline']['executed_frac_of_peak'],3))"
done
timeout 600 python bench.py --precision fp32 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r6h_bench_fp32.json
python -c "import json; j=json.load(open('$O/r6h_bench_fp32.json')); print('fp32 auto', round(j['value'],2), 'x', round(j['ms_per_step'],1), 'ms')"
