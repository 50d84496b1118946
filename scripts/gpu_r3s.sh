#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -q -k "fir" < /dev/null 2>&1 | tail -8
