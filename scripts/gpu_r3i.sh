#!/bin/bash
# round 3 final profiles: rocprofv3 kernel stats of the bench command, HBM-traffic PMC passes, extra bench lines for the record
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-e2e"
rm -rf $O/prof_stats $O/pmc_fetch $O/pmc_write
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- $BENCH < /dev/null > $O/prof_stats.log 2>&1); echo "stats rc=$?"; tail -1 $O/prof_stats.log | cut -c1-300
BENCH1="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-e2e"
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $BENCH1 < /dev/null > $O/pmc_fetch.log 2>&1); echo "fetch rc=$?"
(cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $BENCH1 < /dev/null > $O/pmc_write.log 2>&1); echo "write rc=$?"
find $O/prof_stats -name '*kernel_trace.csv' -size +20M -delete
python profiles/summarize_pmc.py $O/pmc_fetch $O/pmc_write 2 $((8*768*256)) > $O/conv_traffic.json 2>$O/conv_traffic.err; head -c 600 $O/conv_traffic.json; tail -2 $O/conv_traffic.err
SHAPE=1 bash scripts/pmc_conv.sh > $O/pmc_conv.log 2>&1; tail -2 $O/pmc_conv.log
Q="--no-cpu-baseline --no-e2e"
python bench.py $Q --preset flowdec_25s --batch 32 --N 3 --solver midpoint --steps 3 --warmup 1 > $O/bench_cfg3.json 2>/dev/null; python -c "import json; r=json.load(open('$O/bench_cfg3.json')); print('cfg3', round(r['value'],2), round(r['ms_per_step'],1))"
python bench.py $Q --batch 1 --seconds 1 --steps 30 --warmup 5 --conv-algo latency > $O/bench_b1_latency.json 2>/dev/null; python -c "import json; r=json.load(open('$O/bench_b1_latency.json')); print('b1 latency', round(r['value'],2), round(r['ms_per_step'],2))"
python bench.py $Q --precision bf16x3 --steps 3 --warmup 1 > $O/bench_bf16x3.json 2>/dev/null; python -c "import json; r=json.load(open('$O/bench_bf16x3.json')); print('bf16x3', round(r['value'],2), round(r['ms_per_step'],1))"
python bench.py $Q --precision fp32 --steps 2 --warmup 1 > $O/bench_fp32.json 2>/dev/null; python -c "import json; r=json.load(open('$O/bench_fp32.json')); print('fp32', round(r['value'],2), round(r['ms_per_step'],1))"
python bench.py $Q --gpus 2 --backend gloo --share-gpu --steps 3 --warmup 1 > $O/bench_2ranks_1gpu.json 2>$O/bench_2ranks.err; python -c "import json; r=json.load(open('$O/bench_2ranks_1gpu.json')); print('2 ranks / 1 gpu (gloo)', r['n_gpus'], round(r['value'],2), r['per_rank_ms_per_step'], round(r['allgather_ms_per_step'],2))"; tail -2 $O/bench_2ranks.err
