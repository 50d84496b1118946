#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; rm -f $O/parity_report.txt
timeout 1500 python -m pytest tests -m gpu -q --durations=15 ${PYTEST_ARGS:--x} < /dev/null > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -40 $O/pytest_gpu.log
