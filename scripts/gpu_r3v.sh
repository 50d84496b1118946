#!/bin/bash
# cfg 3 (FlowDec-25s preset, 32 x 2 s, midpoint N = 3) kernel trace -> step breakdown
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; rm -rf $O/prof_cfg3
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/prof_cfg3 -- python $GRAFT_REPO_ROOT/bench.py --preset flowdec_25s --batch 32 --N 3 --solver midpoint --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-e2e < /dev/null > $O/prof_cfg3.log 2>&1); echo rc=$?
f=$(ls $O/prof_cfg3/*/*kernel_trace.csv | head -1)
python scripts/step_breakdown.py $f 30 > $O/cfg3_breakdown.txt; head -32 $O/cfg3_breakdown.txt
find $O/prof_cfg3 -name '*kernel_trace.csv' -size +20M -delete
