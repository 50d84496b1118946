#!/bin/bash
# SQ / GRBM counters of the codec's matrix-core kernel (conv1d_mfma_kernel) inside scripts/ndac_bench.py, per launch shape
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
CMD="python $GRAFT_REPO_ROOT/scripts/ndac_bench.py --iters 1"
rm -rf $O/pmcnd1 $O/pmcnd2
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmcnd1 -- $CMD < /dev/null > $O/pmcnd1.log 2>&1); echo rc=$?
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmcnd2 -- $CMD < /dev/null > $O/pmcnd2.log 2>&1); echo rc=$?
python - <<'PY'
import csv, glob, collections, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out")
m = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
def key(r):
    n = r["Kernel_Name"]
    if "conv1d_mfma_kernel" not in n: return None
    t = n[n.index("conv1d_mfma_kernel"):].split("(")[0]
    return (t, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""))
for d in (f"{O}/pmcnd1", f"{O}/pmcnd2"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = key(r)
            if k: m[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = key(r)
            if k: dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for k in sorted(m, key=lambda k: -sum(dur.get(k, [0]))):
    a = {c: sum(v) / len(v) for c, v in m[k].items()}
    if "GRBM_GUI_ACTIVE" not in a or "SQ_WAVE_CYCLES" not in a or not dur.get(k): continue
    cyc = a["GRBM_GUI_ACTIVE"] / 8; us = sum(dur[k]) / len(dur[k]) / 1e3; w = a["SQ_WAVE_CYCLES"]; mf = max(a["SQ_INSTS_MFMA"], 1)
    print(f"{k[0]:34s} grid {k[1]:>7s} x {k[2]:>3s} n {len(dur[k]):3d} {us:7.1f} us clock {cyc / us / 1e3:4.2f} GHz | MFMA busy {a['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024) * 100:4.1f} % | "
          f"wave: active {a['SQ_ACTIVE_INST_ANY'] / w * 100:2.0f} % wait {a['SQ_WAIT_ANY'] / w * 100:2.0f} % | per MFMA: VALU {a['SQ_INSTS_VALU'] / mf:.2f} LDS {a['SQ_INSTS_LDS'] / mf:.2f} VMEM {a['SQ_INSTS_VMEM'] / mf:.2f}")
PY
