#!/usr/bin/env python
"""Per-kernel time per solver step from a `rocprofv3 --kernel-trace --stats --output-format csv` directory (the *_kernel_stats.csv or,
if absent, the *_kernel_trace.csv).  usage: summarize_kernel_stats.py DIR STEPS_WITH_RECORDS"""
import collections
import csv
import glob
import sys


def main(d, steps):
    agg = collections.defaultdict(lambda: [0, 0.0])
    files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    if files:
        for f in files:
            for r in csv.DictReader(open(f)):
                a = agg[r["Kernel_Name"]]
                a[0] += 1
                a[1] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6
    else:
        for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                a = agg[r["Name"]]
                a[0] += int(r["Calls"])
                a[1] += float(r["TotalDurationNs"]) / 1e6
    tot = sum(v[1] for v in agg.values())
    print(f"# {steps} steps with kernel records; listed kernels {tot / steps:.2f} ms per step")
    conv = fir = once = 0.0
    nconv = nstep = 0
    ONCE = ("__amd_rocclr_copyBuffer", "pack_weights_kernel", "wino_pack_kernel", "wino4_pack", "wino4_scale", "wino_scale", "at::native", "fillBuffer")   # model load / weight packing / torch's input generation
    for k, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        short = k.replace("_ZN12_GLOBAL__N_1", "").replace("(anonymous namespace)::", "").replace("void ", "")[:84]
        if any(o in k for o in ONCE):
            once += ms
        else:
            nstep += n
        if ms / steps >= 0.01:
            print(f"{short:84s} calls/step {n / steps:7.1f}  ms/step {ms / steps:8.3f}  avg {1e3 * ms / n:8.1f} us  {100 * ms / tot:5.2f} %")
        if "conv_mfma" in k or "conv_wino" in k or "conv_head" in k:
            conv += ms; nconv += n
        if "fir_" in k:
            fir += ms
    print(f"# conv kernels: {nconv / steps:.0f} launches per step, {conv / steps:.2f} ms per step; FIR {fir / steps:.2f} ms; everything else "
          f"{(tot - conv - fir - once) / steps:.2f} ms per step (+ {once:.2f} ms ONCE per run: model load, weight packing, input generation -- "
          f"the copyBuffer / pack rows above are those, divided by the step count like every row); launches per step {nstep / steps:.0f}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]))
