#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q < /dev/null 2>&1 | tail -3
timeout 300 python bench.py --no-cpu-baseline < /dev/null 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
O=$GRAFT_REPO_ROOT/gpurun_out; rm -rf $O/prof_stats
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline < /dev/null > $O/prof_stats.log 2>&1)
rm -f $O/prof_stats/*/*kernel_trace.csv
head -14 $O/prof_stats/*/*kernel_stats.csv | cut -c1-150
