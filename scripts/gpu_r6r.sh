#!/bin/bash
# round 6, call R: after the float32 pyramid-head kernel (conv_headf.hip) -- full GPU parity suite, smoke, cfg 2 line, the fp32 lines (cfg 2 shape,
# cfg 5 shard), fp32 kernel statistics, head timing in both precisions
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; rm -f $O/parity_report.txt
timeout 1500 python -m pytest tests -m gpu -q < /dev/null > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
timeout 400 python bench.py --steps 20 --warmup 5 < /dev/null > $O/r6r_bench_cfg2.json 2> $O/bench_cfg2.err; cut -c1-200 $O/r6r_bench_cfg2.json
timeout 600 python bench.py --precision fp32 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r6r_bench_fp32.json
python -c "import json; j=json.load(open('$O/r6r_bench_fp32.json')); print('fp32', round(j['value'],2), 'x', round(j['ms_per_step'],1), 'ms frac', round(j['roofline']['frac'],3), 'executed', round(j['roofline']['executed_frac_of_peak'],3))"
timeout 900 python bench.py --config cfg5 --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r6r_bench_cfg5.json
python -c "import json; j=json.load(open('$O/r6r_bench_cfg5.json')); print('cfg5 shard', round(j['value'],2), 'x', round(j['ms_per_step'],1), 'ms frac', round(j['roofline']['frac'],3))"
rm -rf $O/prof_fp32
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fp32 -- python $GRAFT_REPO_ROOT/bench.py --precision fp32 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-e2e --no-calibration < /dev/null > $O/prof_fp32.log 2>&1); echo "fp32 stats rc=$?"
python profiles/summarize_kernel_stats.py $O/prof_fp32 3 > $O/r6r_fp32_kernel_stats.txt 2>&1; head -9 $O/r6r_fp32_kernel_stats.txt | cut -c1-170
find $O/prof_fp32 -name '*.csv' -size +20M -delete
{ python scripts/head_timing.py; HEAD_DT=fp32 python scripts/head_timing.py; } 2>&1 | grep -v amdgpu.ids > $O/r6r_head_timing.txt; cat $O/r6r_head_timing.txt
