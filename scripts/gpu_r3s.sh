#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_ndac.py -m gpu -q -k "full" --durations=3 < /dev/null 2>&1 | tail -12
