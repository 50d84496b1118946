#!/bin/bash
# One gpurun call: full GPU parity suite, smoke, bench line, rocprofv3 kernel stats and the two HBM-traffic PMC passes.
#   gpurun --timeout 1800 -- 'bash scripts/gpu_round.sh'
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O; rm -f $O/parity_report.txt
timeout 900 python -m pytest tests -m gpu -q < /dev/null > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 400 python bench.py < /dev/null > $O/bench_cfg2.json 2> $O/bench_cfg2.err; tail -c 2500 $O/bench_cfg2.json
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline"
rm -rf $O/prof_stats $O/pmc_fetch $O/pmc_write
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- $BENCH < /dev/null > $O/prof_stats.log 2>&1); echo "stats rc=$?"
BENCH1="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-roofline"
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $BENCH1 < /dev/null > $O/pmc_fetch.log 2>&1); echo "fetch rc=$?"
(cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $BENCH1 < /dev/null > $O/pmc_write.log 2>&1); echo "write rc=$?"
# keep only the small summaries (the per-dispatch traces can be large)
find $O/prof_stats -name '*kernel_trace.csv' -size +20M -delete
du -sh $O/prof_stats $O/pmc_fetch $O/pmc_write 2>/dev/null
