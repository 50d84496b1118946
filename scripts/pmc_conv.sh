#!/bin/bash
# SQ counters of the conv kernel on one L0 shape (two passes of <= 8 SQ counters)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; SHAPE=${SHAPE:-1}
CMD="python $GRAFT_REPO_ROOT/scripts/conv_bench.py --B 8 --iters 3 --only $SHAPE $BENCH_ARGS"   # e.g. BENCH_ARGS="--dtype fp32 --operands x3"
rm -rf $O/pmc_sq1 $O/pmc_sq2
(cd /tmp && timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_sq1 -- $CMD < /dev/null > $O/pmc_sq1.log 2>&1); echo rc=$?
(cd /tmp && timeout 200 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq2 -- $CMD < /dev/null > $O/pmc_sq2.log 2>&1); echo rc=$?
tail -3 $O/pmc_sq2.log
