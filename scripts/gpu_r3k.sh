#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_ndac.py -m gpu -q < /dev/null 2>&1 | tail -5
timeout 300 python scripts/ndac_bench.py > $O/ndac_bench.json 2> $O/ndac_bench.err; cat $O/ndac_bench.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash scripts/gpu_r3m.sh
