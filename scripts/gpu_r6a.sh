#!/bin/bash
# round 6, call A: the new ragged-batch tests and the changed Winograd weight header first (fast feedback), then the full GPU suite,
# smoke, the box probe alone, and the cfg 2 bench line with `box_calibration` / `power` / `devices`.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; rm -f $O/parity_report.txt
timeout 900 python -m pytest tests/test_hip_ragged.py -m gpu -q -x < /dev/null > $O/r6a_ragged.log 2>&1; echo "ragged rc=$?" >> $O/r6a_ragged.log; tail -25 $O/r6a_ragged.log
timeout 600 python -m pytest tests/test_hip_configs.py -m gpu -q -x -k "weight_range or winograd4 or winograd" < /dev/null > $O/r6a_wino.log 2>&1; echo "wino rc=$?" >> $O/r6a_wino.log; tail -8 $O/r6a_wino.log
timeout 120 python -m flowdec_amd.boxprobe > $O/r6a_boxprobe.json 2> $O/r6a_boxprobe.err; cat $O/r6a_boxprobe.json; tail -3 $O/r6a_boxprobe.err
ls /sys/class/drm/ 2>/dev/null | head -20; ls /sys/class/drm/card*/device/hwmon/*/ 2>/dev/null | head -40
timeout 400 python bench.py --steps 10 --warmup 3 < /dev/null > $O/r6a_bench_cfg2.json 2> $O/r6a_bench_cfg2.err; python - <<'PY'
import json,os
try:
    j=json.load(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r6a_bench_cfg2.json'))
    print('cfg2', round(j['value'],2), 'x', round(j['ms_per_step'],2), 'ms; calib', j.get('box_calibration'), 'power', j.get('power'), 'vpc', j.get('value_per_calibration'), 'devices', j.get('devices'))
except Exception as e: print('bench parse failed', e)
PY
tail -5 $O/r6a_bench_cfg2.err
timeout 1500 python -m pytest tests -m gpu -q -x < /dev/null > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -8 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
