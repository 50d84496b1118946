#!/bin/bash
# round 3, call G: the whole GPU suite (NDAC, dist, RE opt-in, hygiene) + smoke + default bench line
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; rm -f $O/parity_report.txt
timeout 1500 python -m pytest tests -m gpu -q --durations=8 < /dev/null > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -25 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
timeout 500 python bench.py < /dev/null > $O/bench_cfg2.json 2> $O/bench_cfg2.err; python -c "
import json; r=json.load(open('$O/bench_cfg2.json')); print('value', round(r['value'],2), 'ms', round(r['ms_per_step'],2), 'e2e', round(r['e2e']['value'],2), 'conv frac', round(r['roofline']['frac'],4), 'cpu', r['cpu_baseline']['value'], r['cpu_baseline']['cores'])"
tail -2 $O/bench_cfg2.err
