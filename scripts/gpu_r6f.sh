#!/bin/bash
# round 6, call F: kernel stats of the fp32 mode with the float32 F(4,3) kernel in the schedule (what is left on the direct f32 kernel?)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; rm -rf $O/prof_fp32
BENCH="python $GRAFT_REPO_ROOT/bench.py --precision fp32 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-e2e --no-calibration"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fp32 -- $BENCH < /dev/null > $O/prof_fp32.log 2>&1); echo "stats rc=$?"
python profiles/summarize_kernel_stats.py $O/prof_fp32 3 > $O/r6f_fp32_kernel_stats.txt 2>&1; head -30 $O/r6f_fp32_kernel_stats.txt | cut -c1-200
f=$(ls $O/prof_fp32/*/*kernel_trace.csv | head -1); python scripts/step_breakdown.py $f 40 > $O/r6f_fp32_step_breakdown.txt 2>&1; head -45 $O/r6f_fp32_step_breakdown.txt
find $O/prof_fp32 -name '*.csv' -size +20M -delete
