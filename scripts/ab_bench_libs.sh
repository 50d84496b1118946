#!/bin/bash
# ab_bench_libs.sh ROUNDS NAME...: interleaved cfg 2 bench lines (ms per step) of variant libraries flowdec_amd/variants/libflowdec_NAME.so
# ("hip" = the product library) in ONE gpurun call -- boxes of the pool differ by +-2.5 %, an A/B is only valid inside a call
cd $GRAFT_REPO_ROOT
R=$1; shift
for i in $(seq $R); do for v in "$@"; do
  L=$GRAFT_REPO_ROOT/flowdec_amd/variants/libflowdec_$v.so; [ $v = hip ] && L=$GRAFT_REPO_ROOT/flowdec_amd/libflowdec_hip.so
  FLOWDEC_HIP_LIB=$L timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-e2e 2>/dev/null | grep '^{' | python -c "import json,sys; j=json.loads(sys.stdin.readline()); print('$v', round(j['ms_per_step'],2), 'ms', round(j['value'],1), 'x')"
done; done
