#!/bin/bash
# round 5, call A: full GPU parity suite (new: G21 at cfg 2's image size, weight-range sweeps, derived nf = 8 tolerances), smoke, the bench
# line with in-call noise, in-call A/B of the static-priority variants of conv_wino4 (scripts/build_w4prod.sh), kernel stats.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; rm -f $O/parity_report.txt
timeout 1100 python -m pytest tests -m gpu -q -x < /dev/null > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -6 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
timeout 400 python bench.py --steps 10 --warmup 3 < /dev/null > $O/bench_cfg2.json 2> $O/bench_cfg2.err; tail -c 1500 $O/bench_cfg2.json | cut -c1-1500
bash scripts/ab_bench_libs.sh 2 hip prio1 prio0 2>&1 | tee $O/r5a_ab_prio.txt
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-e2e"
rm -rf $O/prof_stats
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- $BENCH < /dev/null > $O/prof_stats.log 2>&1); echo "stats rc=$?"
python profiles/summarize_kernel_stats.py $O/prof_stats 7 > $O/r5a_kernel_stats.txt 2>&1; head -40 $O/r5a_kernel_stats.txt
find $O/prof_stats -name '*kernel_trace.csv' -size +20M -delete
