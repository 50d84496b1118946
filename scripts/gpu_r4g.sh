#!/bin/bash
# round 4, call G: full GPU suite (new tests included) + smoke + cfg 2 line + cfg 3 lines (N = 3 and N = 6) + kernel stats + HBM traffic passes
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; rm -f $O/parity_report.txt
timeout 1500 python -m pytest tests -m gpu -q --durations=8 < /dev/null > $O/r4g_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r4g_pytest.log; tail -14 $O/r4g_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py 2> $O/r4g_bench_cfg2.err | tail -1 > $O/r4g_bench_cfg2.json; cut -c1-300 $O/r4g_bench_cfg2.json
for N in 3 6; do
  timeout 600 python bench.py --preset flowdec_25s --batch 32 --solver midpoint --N $N --steps 3 --warmup 1 --no-cpu-baseline 2> $O/r4g_bench_cfg3_N$N.err | tail -1 > $O/r4g_bench_cfg3_N$N.json; cut -c1-300 $O/r4g_bench_cfg3_N$N.json
done
timeout 600 python bench.py --batch 32 --solver midpoint --N 3 --steps 3 --warmup 1 --no-cpu-baseline 2> /dev/null | tail -1 > $O/r4g_bench_cfg4_shard.json; cut -c1-300 $O/r4g_bench_cfg4_shard.json
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-e2e"
rm -rf $O/prof_stats $O/pmc_fetch $O/pmc_write
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- $BENCH < /dev/null > $O/prof_stats.log 2>&1); echo "stats rc=$?"
python profiles/summarize_kernel_stats.py $O/prof_stats 7 > $O/r4g_kernel_stats.txt; head -30 $O/r4g_kernel_stats.txt
BENCH1="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-e2e"
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $BENCH1 < /dev/null > $O/pmc_fetch.log 2>&1); echo "fetch rc=$?"
(cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $BENCH1 < /dev/null > $O/pmc_write.log 2>&1); echo "write rc=$?"
python profiles/summarize_pmc.py $O/pmc_fetch $O/pmc_write 2 1572864 > $O/r4g_conv_traffic.json; head -c 1500 $O/r4g_conv_traffic.json
find $O/prof_stats $O/pmc_fetch $O/pmc_write -name '*.csv' -size +20M -delete
du -sh $O/prof_stats $O/pmc_fetch $O/pmc_write 2>/dev/null
