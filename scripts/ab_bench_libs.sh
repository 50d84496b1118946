#!/bin/bash
# ab_bench_libs.sh ROUNDS NAME...: interleaved cfg 2 bench lines (ms per step) of variant libraries flowdec_amd/variants/libflowdec_NAME.so
# ("hip" = the product library) in ONE gpurun call -- boxes of the pool differ by +-2.5 %, an A/B is only valid inside a call.
# Round 6: POWER-AWARE -- every line carries the mean shader clock and package power of its timed region (bench.py `power`, sampled by
# flowdec_amd/boxprobe.py) and the box calibration: the chip runs this workload at its power limit, so a variant that removes idle cycles
# can be paid back by a lower clock, and one that corrupts its data can look faster (MEASUREMENTS R4.2).  ms x MHz = cycles per step.
cd $GRAFT_REPO_ROOT
R=$1; shift
for i in $(seq $R); do for v in "$@"; do
  L=$GRAFT_REPO_ROOT/flowdec_amd/variants/libflowdec_$v.so; [ $v = hip ] && L=$GRAFT_REPO_ROOT/flowdec_amd/libflowdec_hip.so
  FLOWDEC_HIP_LIB=$L timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-e2e 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); p=j.get('power') or {}; c=j.get('box_calibration') or {}
f=lambda x,n=0: ('%.*f' % (n, x)) if isinstance(x,(int,float)) else 'n/a'
mc = j['ms_per_step']*p['sclk_mhz']/1e3 if p.get('sclk_mhz') else None
print('$v', round(j['ms_per_step'],2), 'ms', round(j['value'],1), 'x | sclk', f(p.get('sclk_mhz')), 'MHz', f(p.get('power_w')), 'W (', p.get('samples'), 'samples ) | Mcycles/step', f(mc,1), '| calib', f(c.get('mfma_tflops')), 'TF @', f(c.get('sclk_mhz')), 'MHz', f(c.get('power_w')), 'W')"
done; done
