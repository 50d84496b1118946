#!/usr/bin/env python
"""Turn the two rocprofv3 PMC passes (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, counter_collection.csv) of a bench.py run
into per-kernel HBM traffic, with the unit / gfx950 corrections of /opt/skills/guides/MI355X_MICROARCH.md (HBM section):
both counters are in KiB; FETCH_SIZE reports half of the bytes of coalesced streaming reads on gfx950 -> doubled.
The correction is calibrated on kernels of this run whose byte counts are known exactly (see `calibration` in the output).

    python profiles/summarize_pmc.py gpurun_out/pmc_fetch gpurun_out/pmc_write <steps-in-run> [B*F*T_pad] > profiles/rNN_conv_traffic.json
"""
import collections
import csv
import glob
import json
import sys


def load(d):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            for pre in ("_ZN12_GLOBAL__N_1", "(anonymous namespace)::"):
                name = name.replace(pre, "")
            key = name.split("(")[0].lstrip("0123456789")
            if key.startswith("void "):
                key = key[5:]
            key = key[:60]
            agg[key][0] += 1
            agg[key][1] += float(r["Counter_Value"]) * 1024.0
    return agg


def main(fetch_dir, write_dir, steps, pixels=0):
    fe, wr = load(fetch_dir), load(write_dir)
    out = {"units": "bytes", "steps_in_run": steps, "corrections": "KiB -> bytes; FETCH_SIZE x2 (gfx950 half-count of wide coalesced reads)", "kernels": {}}
    conv = dict(launches=0, fetch_raw=0.0, write=0.0)
    fir = dict(launches=0, fetch_raw=0.0, write=0.0)
    for k in sorted(set(fe) | set(wr), key=lambda k: -(fe.get(k, [0, 0])[1] + wr.get(k, [0, 0])[1])):
        n = max(fe.get(k, [0, 0])[0], wr.get(k, [0, 0])[0])
        if not k or n == 0:
            continue
        f, w = fe.get(k, [0, 0.0])[1], wr.get(k, [0, 0.0])[1]
        out["kernels"][k] = dict(launches=n, fetch_raw_per_launch=f / n, fetch_corrected_per_launch=2 * f / n, write_per_launch=w / n)
        if k.startswith(("conv_mfma_kernel", "conv_wino_kernel", "conv_wino4_kernel", "conv_head_kernel")):
            conv["launches"] += n; conv["fetch_raw"] += f; conv["write"] += w
        if k.startswith(("fir_up_kernel", "fir_down_kernel", "fir_down_march_kernel")):
            fir["launches"] += n; fir["fetch_raw"] += f; fir["write"] += w
    n = conv["launches"]
    out["conv_mfma_kernel"] = dict(launches=n, launches_per_step=n // steps, fetch_corrected_per_launch=2 * conv["fetch_raw"] / n,
                                   write_per_launch=conv["write"] / n, traffic_per_launch=(2 * conv["fetch_raw"] + conv["write"]) / n,
                                   traffic_per_step=(2 * conv["fetch_raw"] + conv["write"]) / steps)
    n = max(fir["launches"], 1)
    out["fir_kernels"] = dict(launches=fir["launches"], launches_per_step=fir["launches"] // steps, fetch_corrected_per_launch=2 * fir["fetch_raw"] / n,
                              write_per_launch=fir["write"] / n, traffic_per_launch=(2 * fir["fetch_raw"] + fir["write"]) / n,
                              traffic_per_step=(2 * fir["fetch_raw"] + fir["write"]) / steps)
    if pixels:  # kernels with exactly known traffic: pack_input reads two complex64 planes and writes 8 bf16 channels per pixel
        for k, v in out["kernels"].items():
            if k.startswith("pack_input_kernel"):
                out["calibration"] = dict(kernel="pack_input_kernel", expected_read=16.0 * pixels, measured_read_corrected=v["fetch_corrected_per_launch"],
                                          expected_write=16.0 * pixels, measured_write=v["write_per_launch"])
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) > 4 else 0)
