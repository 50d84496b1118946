#!/bin/bash
# round 4, call F: new F(4,3) tests + kernel trace of the cfg 2 bench (per-kernel time per step)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_configs.py -q -x -m gpu -k "winograd4 or shortcut_dynamic or residual_stream" 2>&1 | tail -3
rm -rf $O/prof_stats
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-e2e < /dev/null > $O/prof_stats.log 2>&1); echo "stats rc=$?"
find $O/prof_stats -name '*kernel_trace.csv' -size +20M -delete
python profiles/summarize_kernel_stats.py $O/prof_stats 7 | tee $O/r4f_kernel_stats.txt | head -45
