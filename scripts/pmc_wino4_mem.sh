#!/bin/bash
# vector-memory path counters of the F(4,3) kernel (TA / TCP / TCC) on one shape
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; SHAPE=${SHAPE:-1}
CMD="python $GRAFT_REPO_ROOT/scripts/wino4_check.py --B 8 --iters 3 --rounds 1 --only $SHAPE --no-parity"
rm -rf $O/pmc4m_*
i=0
for set in "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN2_sum" "TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_BUSY_avr" "SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  (cd /tmp && timeout 200 rocprofv3 --pmc $set --output-format csv -d $O/pmc4m_$i -- $CMD < /dev/null > $O/pmc4m_$i.log 2>&1); echo "set $i rc=$?"; grep -i "error\|invalid\|not supported" $O/pmc4m_$i.log | head -2
done
python - <<'PY'
import csv, glob, collections, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/pmc4m_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_" in r["Kernel_Name"] and "pack" not in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    print(k)
    for n in sorted(c): print(f"  {n:36s} {sum(c[n]) / len(c[n]):16.0f}")
PY
