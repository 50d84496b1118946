#!/bin/bash
# round 6, call Q: the COMPLETE evidence set of the final code (same recipe as call M; after the last changes to conv_wino44f.hip and conv_head.hip) -- HBM-traffic PMC passes first (the bench line quotes them), full GPU parity suite,
# smoke, the bench line (driver-style), kernel stats, SQ counters of the F(4,3) kernel, the other BASELINE configurations and modes
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; rm -f $O/parity_report.txt
BENCH1="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-e2e --no-calibration"
rm -rf $O/prof_stats $O/pmc_fetch $O/pmc_write
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $BENCH1 < /dev/null > $O/pmc_fetch.log 2>&1); echo "fetch rc=$?"
(cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $BENCH1 < /dev/null > $O/pmc_write.log 2>&1); echo "write rc=$?"
python profiles/summarize_pmc.py $O/pmc_fetch $O/pmc_write 2 $((8*768*256)) > $O/r6q_conv_traffic.json 2> $O/r6q_conv_traffic.err; grep -A6 '"conv_mfma_kernel"' $O/r6q_conv_traffic.json | head -8
cp $O/r6q_conv_traffic.json profiles/r06_conv_traffic.json
timeout 1500 python -m pytest tests -m gpu -q < /dev/null > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
timeout 400 python bench.py --steps 20 --warmup 5 < /dev/null > $O/r6q_bench_cfg2.json 2> $O/bench_cfg2.err; cut -c1-400 $O/r6q_bench_cfg2.json
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-e2e --no-calibration"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- $BENCH < /dev/null > $O/prof_stats.log 2>&1); echo "stats rc=$?"
python profiles/summarize_kernel_stats.py $O/prof_stats 7 > $O/r6q_kernel_stats.txt 2>&1; tail -2 $O/r6q_kernel_stats.txt
SHAPE=1 bash scripts/pmc_wino4.sh 2>&1 | grep -v "^rc=" > $O/r6q_wino4_pmc.txt; grep -- "->" $O/r6q_wino4_pmc.txt
for cfg in cfg3 cfg3_nfe6 cfg4 cfg4_nfe6; do
  timeout 600 python bench.py --config $cfg --steps 3 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r6q_bench_$cfg.json
  python -c "import json; j=json.load(open('$O/r6q_bench_$cfg.json')); print('$cfg', round(j['value'],1), 'x', round(j['ms_per_step'],1), 'ms frac', round(j['roofline']['frac'],3), 'nfe', j['config']['nfe'])"
done
for sec in 1 2; do
  timeout 300 python bench.py --batch 1 --seconds $sec --conv-algo latency --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r6q_bench_b1_${sec}s_latency.json
  python -c "import json; j=json.load(open('$O/r6q_bench_b1_${sec}s_latency.json')); print('b1 latency', $sec, 's:', round(j['value'],2), 'x', round(j['ms_per_step'],3), 'ms')"
done
for prec in bf16x3 fp32; do
  timeout 600 python bench.py --precision $prec --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r6q_bench_$prec.json
  python -c "import json; j=json.load(open('$O/r6q_bench_$prec.json')); print('$prec', round(j['value'],1), 'x', round(j['ms_per_step'],1), 'ms frac', round(j['roofline']['frac'],3), 'executed', round(j['roofline']['executed_frac_of_peak'],3))"
done
timeout 600 python bench.py --precision fp32 --conv-algo direct --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r6q_bench_fp32_direct.json
python -c "import json; j=json.load(open('$O/r6q_bench_fp32_direct.json')); print('fp32 direct', round(j['value'],1), 'x', round(j['ms_per_step'],1), 'ms frac', round(j['roofline']['frac'],3))"
timeout 900 python bench.py --config cfg5 --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r6q_bench_cfg5.json
python -c "import json; j=json.load(open('$O/r6q_bench_cfg5.json')); print('cfg5 shard', round(j['value'],2), 'x', round(j['ms_per_step'],1), 'ms frac', round(j['roofline']['frac'],3))"
rm -rf $O/prof_fp32
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fp32 -- python $GRAFT_REPO_ROOT/bench.py --precision fp32 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-e2e --no-calibration < /dev/null > $O/prof_fp32.log 2>&1); echo "fp32 stats rc=$?"
python profiles/summarize_kernel_stats.py $O/prof_fp32 3 > $O/r6q_fp32_kernel_stats.txt 2>&1; head -12 $O/r6q_fp32_kernel_stats.txt | cut -c1-170
find $O/prof_stats $O/pmc_fetch $O/pmc_write $O/prof_fp32 -name '*.csv' -size +20M -delete
FLOWDEC_HIP_LIB=$GRAFT_REPO_ROOT/flowdec_amd/variants/libflowdec_t2f.so timeout 600 python scripts/wino44f_timing2.py 2>&1 | grep -v amdgpu.ids > $O/r6q_w44_timing.txt; cut -c1-150 $O/r6q_w44_timing.txt
