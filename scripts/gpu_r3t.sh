#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
Q="--no-cpu-baseline --no-e2e --steps 10 --warmup 3"
for i in 1 2; do
python bench.py $Q > $O/b_side.json 2>/dev/null; python -c "import json; r=json.load(open('$O/b_side.json')); print('side stream   ', round(r['value'],2), round(r['ms_per_step'],2), round(r['roofline']['conv_ms_per_step'],2))"
python bench.py $Q --no-side-stream > $O/b_noside.json 2>/dev/null; python -c "import json; r=json.load(open('$O/b_noside.json')); print('no side stream', round(r['value'],2), round(r['ms_per_step'],2), round(r['roofline']['conv_ms_per_step'],2))"
done
