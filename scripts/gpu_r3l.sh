#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_model.py tests/test_hip_ndac.py tests/test_cli.py -m gpu -q < /dev/null 2>&1 | tail -5
