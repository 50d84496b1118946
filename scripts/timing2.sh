#!/bin/bash
cd $GRAFT_REPO_ROOT
FLOWDEC_EXTRA_FLAGS="-DFD_TIMING2" python flowdec_amd/build.py --force > /dev/null 2>&1 < /dev/null || echo BUILD FAILED
timeout 200 python scripts/conv_timing2.py < /dev/null 2>&1 | grep -v amdgpu
python flowdec_amd/build.py --force > /dev/null 2>&1 < /dev/null
