#!/bin/bash
# round 6, call K: 2-D float32 Winograd kernel after the rewrite: operator parity, phase split, fp32 bench
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; rm -f $O/parity_report.txt
timeout 900 python -m pytest tests/test_hip_configs.py -m gpu -q -k "winograd4_f32 or fp32_auto" < /dev/null > $O/r6k_w44.log 2>&1; echo "w44 rc=$?" >> $O/r6k_w44.log; tail -5 $O/r6k_w44.log | cut -c1-220
grep "winograd44" $O/parity_report.txt | cut -c1-110
FLOWDEC_HIP_LIB=$GRAFT_REPO_ROOT/flowdec_amd/variants/libflowdec_t2f.so timeout 600 python scripts/wino44f_timing2.py 2>&1 | tee $O/r6k_w44_timing.txt
timeout 600 python bench.py --precision fp32 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r6k_bench_fp32.json
python -c "import json; j=json.load(open('$O/r6k_bench_fp32.json')); print('fp32 cfg2-shape auto', round(j['value'],2), 'x', round(j['ms_per_step'],1), 'ms frac', round(j['roofline']['frac'],3), 'exec', round(j['roofline']['executed_frac_of_peak'],3))"
