#!/bin/bash
# SQ counters (three passes of <= 8) of the three float32 convolution kernels (direct, F(4,3), 2-D F(4x4,3x3)) on the affine 256 -> 256 shape
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; SHAPE=${SHAPE:-1}
CMD="python $GRAFT_REPO_ROOT/scripts/wino44f_pmc_driver.py"
rm -rf $O/pmc44_1 $O/pmc44_2 $O/pmc44_3
(cd /tmp && timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc44_1 -- $CMD < /dev/null > $O/pmc44_1.log 2>&1); echo rc=$?
(cd /tmp && timeout 200 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/pmc44_2 -- $CMD < /dev/null > $O/pmc44_2.log 2>&1); echo rc=$?
(cd /tmp && timeout 200 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM --output-format csv -d $O/pmc44_3 -- $CMD < /dev/null > $O/pmc44_3.log 2>&1); echo rc=$?
python - <<'PY'
import csv, glob, collections, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in (f"{O}/pmc44_1", f"{O}/pmc44_2", f"{O}/pmc44_3"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "conv_" in r["Kernel_Name"] and "pack" not in r["Kernel_Name"]:
                agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    print(k)
    for n in sorted(m): print(f"  {n:28s} {m[n]:16.0f}")
    cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8
    if cyc and "SQ_INSTS_MFMA" in m:
        nm = m["SQ_INSTS_MFMA"]
        print(f"  -> cycles/dispatch {cyc:.0f}; MFMA pipe busy {m['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024) * 100:.1f} %; per MFMA: VALU {m['SQ_INSTS_VALU'] / nm:.2f} SALU {m['SQ_INSTS_SALU'] / nm:.2f} LDS {m['SQ_INSTS_LDS'] / nm:.2f} VMEM {m['SQ_INSTS_VMEM'] / nm:.2f}")
    if "SQ_WAVE_CYCLES" in m:
        w = m["SQ_WAVE_CYCLES"]
        print(f"  -> wave time: active {m['SQ_ACTIVE_INST_ANY'] / w * 100:.0f} % waitcnt/barrier {m['SQ_WAIT_ANY'] / w * 100:.0f} % issue stall {m['SQ_WAIT_INST_ANY'] / w * 100:.0f} % (LDS {m['SQ_WAIT_INST_LDS'] / w * 100:.0f} %); LDS array busy {m['SQ_LDS_IDX_ACTIVE'] / max(m.get('SQ_BUSY_CYCLES', 1), 1) * 100:.1f} % of SQ busy cycles, conflict {m['SQ_LDS_BANK_CONFLICT'] / max(m['SQ_LDS_IDX_ACTIVE'], 1) * 100:.1f} % of LDS cycles")
PY
