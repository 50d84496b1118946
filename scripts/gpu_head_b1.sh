#!/bin/bash
# head-kernel A/B (dedicated C->4 kernel vs the generic BN=32 configuration) + kernel trace of the B = 1 x 1 s latency case
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_ops.py tests/test_hip_model.py -m gpu -q -x -k "conv or ncsnpp or enhance or resblock" < /dev/null > $O/pytest_head.log 2>&1; echo "pytest rc=$?" >> $O/pytest_head.log
tail -5 $O/pytest_head.log
timeout 600 python scripts/ab_conv.py old=flowdec_amd/variants/libflowdec_nohead.so new=flowdec_amd/libflowdec_hip.so --rounds 2 2>&1 | tail -14 | tee $O/ab_head.txt
Q="--no-cpu-baseline --no-roofline --steps 5 --warmup 2"
for L in flowdec_amd/variants/libflowdec_nohead.so flowdec_amd/libflowdec_hip.so; do
  FLOWDEC_HIP_LIB=$GRAFT_REPO_ROOT/$L python bench.py $Q 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('cfg2', '$L', round(r['value'],2), 'x', round(r['ms_per_step'],2), 'ms')"
  FLOWDEC_HIP_LIB=$GRAFT_REPO_ROOT/$L python bench.py $Q --batch 1 --seconds 1 --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('B=1x1s', '$L', round(r['value'],2), 'x', round(r['ms_per_step'],2), 'ms')"
done | tee $O/bench_head.txt
rm -rf $O/prof_b1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b1 -- python $GRAFT_REPO_ROOT/bench.py --batch 1 --seconds 1 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline < /dev/null > $O/prof_b1.log 2>&1); echo "prof rc=$?"
du -sh $O/prof_b1
