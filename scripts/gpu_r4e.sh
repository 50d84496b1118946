#!/bin/bash
# round 4, call E: full GPU suite + smoke + cfg 2 bench line
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x < /dev/null > $O/r4e_pytest.log 2>&1; tail -5 $O/r4e_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/r4e_bench.json | cut -c1-420
