cd $GRAFT_REPO_ROOT
for i in 1 2; do for v in lat0 hip; do
  L=$GRAFT_REPO_ROOT/flowdec_amd/variants/libflowdec_$v.so; [ $v = hip ] && L=$GRAFT_REPO_ROOT/flowdec_amd/libflowdec_hip.so
  for sec in 1 2; do FLOWDEC_HIP_LIB=$L timeout 300 python bench.py --batch 1 --seconds $sec --conv-algo latency --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-e2e 2>/dev/null | grep '^{' | python -c "import json,sys; j=json.loads(sys.stdin.readline()); print('$v', $sec, 's', round(j['ms_per_step'],3), 'ms', round(j['value'],1), 'x')"; done
done; done
