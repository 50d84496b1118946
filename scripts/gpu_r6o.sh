#!/bin/bash
# round 6, call O: last check of the final code (after the per-layer LDS allocation of conv_head.hip): full GPU suite, smoke, bench line
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; rm -f $O/parity_report.txt
timeout 1500 python -m pytest tests -m gpu -q < /dev/null > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
timeout 400 python bench.py --steps 20 --warmup 5 < /dev/null > $O/r6o_bench_cfg2.json 2> $O/bench_cfg2.err; cut -c1-300 $O/r6o_bench_cfg2.json
