#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 500 python bench.py --no-cpu-baseline < /dev/null > $O/bench_cfg2_fir.json 2> $O/bench_cfg2_fir.err; python -c "
import json; r=json.load(open('$O/bench_cfg2_fir.json')); print('value', round(r['value'],2), 'ms', round(r['ms_per_step'],2), 'e2e', round(r['e2e']['value'],2), 'conv frac', round(r['roofline']['frac'],4)); print({k:v for k,v in r.items() if k.startswith('roofline')})"
