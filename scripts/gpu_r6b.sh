#!/bin/bash
# round 6, call B: (1) the F(4,3) kernel's STRUCTURAL variants as timing-only builds (scripts/conv_wino4_r6_ablation.patch, built locally with
# scripts/build_w4var.sh): phase split per workgroup -> profiles/r06_wino4_ablation.txt; (2) the one-clip launch trace grouped by grid size
# (is the <= 96-tile tail worth a persistent kernel?) -> profiles/r06_b1_latency_breakdown.txt.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
: > $O/r6b_wino4_ablation_raw.txt
for v in t2 halfw prod2 a128 a128cs cs noconv halfb t2; do
  echo "== $v" >> $O/r6b_wino4_ablation_raw.txt
  FLOWDEC_HIP_LIB=$GRAFT_REPO_ROOT/flowdec_amd/variants/libflowdec_$v.so timeout 300 python scripts/wino4_timing2.py 2>&1 | grep wino4 >> $O/r6b_wino4_ablation_raw.txt
done
cat $O/r6b_wino4_ablation_raw.txt | cut -c1-200
# one clip, 1 s, latency schedule: kernel trace -> per (kernel, grid) breakdown of one step
rm -rf $O/prof_b1
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/prof_b1 -- python $GRAFT_REPO_ROOT/bench.py --batch 1 --seconds 1 --conv-algo latency --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-e2e --no-calibration < /dev/null > $O/prof_b1.log 2>&1); echo "trace rc=$?"
f=$(ls $O/prof_b1/*/*kernel_trace.csv | head -1)
python scripts/step_breakdown.py $f 60 > $O/r6b_b1_breakdown.txt 2>&1; head -70 $O/r6b_b1_breakdown.txt
find $O/prof_b1 -name '*kernel_trace.csv' -size +20M -delete
timeout 300 python bench.py --batch 1 --seconds 1 --conv-algo latency --steps 20 --warmup 5 --no-cpu-baseline < /dev/null 2>/dev/null | tail -1 > $O/r6b_bench_b1_1s_latency.json; cut -c1-300 $O/r6b_bench_b1_1s_latency.json
# the rest of the GPU suite (call A stopped at the first failure: a test-side magnitude that the F(2,3) kernel never supported)
timeout 1500 python -m pytest tests -m gpu -q -x < /dev/null > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -6 $O/pytest_gpu.log
