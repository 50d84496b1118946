#!/bin/bash
# One gpurun call: full GPU parity suite, bench line, rocprofv3 kernel stats and the two HBM-traffic PMC passes.
#   gpurun --timeout 1500 -- 'bash scripts/gpu_round.sh'
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q < /dev/null > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
timeout 300 python bench.py < /dev/null > $O/bench_cfg2.json 2> $O/bench_cfg2.err; tail -c 1500 $O/bench_cfg2.json
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline"
rm -rf $O/prof_stats $O/pmc_fetch $O/pmc_write
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- $BENCH < /dev/null > $O/prof_stats.log 2>&1); echo "stats rc=$?"
BENCH1="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-roofline"
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $BENCH1 < /dev/null > $O/pmc_fetch.log 2>&1); echo "fetch rc=$?"
(cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $BENCH1 < /dev/null > $O/pmc_write.log 2>&1); echo "write rc=$?"
# keep only the small summaries (the per-dispatch traces can be large)
find $O/prof_stats -name '*kernel_trace.csv' -size +20M -delete
du -sh $O/prof_stats $O/pmc_fetch $O/pmc_write 2>/dev/null
