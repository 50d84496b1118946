#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in "" "-DFD_HALO_AUX=2" "-DFD_NT_STORE" "-DFD_HALO_AUX=2 -DFD_NT_STORE"; do
  echo "=== variant [$v]"
  FLOWDEC_EXTRA_FLAGS="$v" python flowdec_amd/build.py --force > /dev/null 2>&1 < /dev/null || echo BUILD FAILED
  for o in 0 1 2 6; do timeout 120 python scripts/conv_bench.py --B 8 --iters 10 --only $o < /dev/null 2>&1 | grep TFLOP; done
  timeout 300 python bench.py --no-cpu-baseline < /dev/null 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
done
python flowdec_amd/build.py --force > /dev/null 2>&1 < /dev/null
