#!/bin/bash
# SQ / GRBM counters of the direct conv kernel for several builds (FLOWDEC_HIP_LIB) on one shape: cycles vs wall time (DVFS)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; SHAPE=${SHAPE:-1}
for NAME in "$@"; do
export FLOWDEC_HIP_LIB=$GRAFT_REPO_ROOT/flowdec_amd/variants/libflowdec_$NAME.so
CMD="python $GRAFT_REPO_ROOT/scripts/wino_check.py --B 8 --iters 20 --rounds 1 --only $SHAPE --no-parity --algo direct"
rm -rf $O/pmcab_${NAME}1 $O/pmcab_${NAME}2
(cd /tmp && timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmcab_${NAME}1 -- $CMD < /dev/null > $O/pmcab_${NAME}1.log 2>&1)
(cd /tmp && timeout 200 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmcab_${NAME}2 -- $CMD < /dev/null > $O/pmcab_${NAME}2.log 2>&1)
grep "^time" $O/pmcab_${NAME}2.log
done
python - "$@" <<'PY'
import csv, glob, collections, os, sys
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out")
for name in sys.argv[1:]:
    m = collections.defaultdict(list); dur = []
    for d in (f"{O}/pmcab_{name}1", f"{O}/pmcab_{name}2"):
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "conv_mfma_kernel" in r["Kernel_Name"]:
                    m[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "conv_mfma_kernel" in r["Kernel_Name"]:
                    dur.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    a = {k: sum(v) / len(v) for k, v in m.items()}
    cyc = a["GRBM_GUI_ACTIVE"] / 8
    us = sum(dur) / len(dur) / 1e3 if dur else float("nan")
    w = a["SQ_WAVE_CYCLES"]
    print(f"{name:8s} cycles/dispatch {cyc:9.0f}  kernel {us:8.1f} us  -> clock {cyc / us / 1e3:5.2f} GHz | MFMA busy {a['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024) * 100:4.1f} % | "
          f"wave: active {a['SQ_ACTIVE_INST_ANY'] / w * 100:2.0f} % wait {a['SQ_WAIT_ANY'] / w * 100:2.0f} % issue-stall {a['SQ_WAIT_INST_ANY'] / w * 100:2.0f} % | "
          f"per MFMA: VALU {a['SQ_INSTS_VALU'] / a['SQ_INSTS_MFMA']:.2f} SALU {a['SQ_INSTS_SALU'] / a['SQ_INSTS_MFMA']:.2f} LDS {a['SQ_INSTS_LDS'] / a['SQ_INSTS_MFMA']:.2f}")
PY
