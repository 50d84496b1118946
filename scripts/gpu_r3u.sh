#!/bin/bash
# one-clip (B = 1 x 1 s, conv_algo latency) kernel trace -> step breakdown
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; rm -rf $O/prof_b1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_b1 -- python $GRAFT_REPO_ROOT/bench.py --batch 1 --seconds 1 --steps 5 --warmup 3 --conv-algo latency --no-cpu-baseline --no-roofline --no-e2e < /dev/null > $O/prof_b1.log 2>&1); echo rc=$?
f=$(ls $O/prof_b1/*/*kernel_trace.csv | head -1)
python scripts/step_breakdown.py $f 48 > $O/b1_latency_breakdown.txt; head -30 $O/b1_latency_breakdown.txt
find $O/prof_b1 -name '*kernel_trace.csv' -size +20M -delete
