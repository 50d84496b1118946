#!/bin/bash
# round 3, call A: full GPU suite (new: dist tests, dynamic-range cases, flag validation, threads), bench line with e2e /
# roofline_stft / torch-CPU baseline, cfg 5 with dopri5 at 1e-4 and 1e-3.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O; rm -f $O/parity_report.txt
timeout 1200 python -m pytest tests -m gpu -q --durations=12 < /dev/null > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -30 $O/pytest_gpu.log
timeout 500 python bench.py < /dev/null > $O/bench_cfg2.json 2> $O/bench_cfg2.err; tail -c 3500 $O/bench_cfg2.json; tail -3 $O/bench_cfg2.err
timeout 600 python scripts/bench_cfg5_dopri5.py --tol 1e-4 --out $O/bench_cfg5_dopri5.json < /dev/null > $O/cfg5.log 2>&1; tail -5 $O/cfg5.log
timeout 400 python scripts/bench_cfg5_dopri5.py --tol 1e-3 --out $O/bench_cfg5_dopri5_tol1e-3.json < /dev/null > $O/cfg5b.log 2>&1; tail -3 $O/cfg5b.log
