#!/bin/bash
# precision='mixed' (f32 residual stream, bf16 operands): parity + speed next to the bf16 and fp32 modes
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_configs.py -m gpu -q -x -k "mixed or bf16x3" < /dev/null > $O/pytest_mixed.log 2>&1; echo "pytest rc=$?" >> $O/pytest_mixed.log
tail -15 $O/pytest_mixed.log
grep -E 'mixed|bf16x3' $O/parity_report.txt
for P in bf16x3; do timeout 400 python bench.py --precision $P --steps 3 --warmup 2 --no-cpu-baseline < /dev/null > $O/bench_$P.json 2> $O/bench_$P.err; python -c "
import json; d=json.load(open('$O/bench_$P.json')); r=d['roofline']; print('$P', round(d['value'],2), 'x', round(d['ms_per_step'],2), 'ms conv', round(r['achieved'],1), 'TF', round(r['avg_launch_ms']*1e3,1), 'us/launch', d['roofline_hbm']['ms_per_step'])"; done
