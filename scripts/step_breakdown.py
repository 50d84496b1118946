#!/usr/bin/env python
"""Per-kernel breakdown of ONE solver step from a rocprofv3 --kernel-trace CSV (bench.py run): kernels between the last two
STFT frame_kernel launches, grouped by (kernel, workgroups).  python scripts/step_breakdown.py <..._kernel_trace.csv> [top]"""
import collections
import csv
import sys


def main(path, top=40):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    idx = [i for i, r in enumerate(rows) if 'frame_kernel' in r['Kernel_Name']]
    step = rows[idx[-2]:idx[-1]]
    t0, t1 = int(step[0]['Start_Timestamp']), int(step[-1]['End_Timestamp'])
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in step:
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        nm = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
        if nm.startswith('_ZN12_GLOBAL__N_1'):
            nm = nm[len('_ZN12_GLOBAL__N_1'):].lstrip('0123456789')
        g = int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X']))
        agg[(nm[:56], g)][0] += 1
        agg[(nm[:56], g)][1] += d
    busy = sum(v[1] for v in agg.values())
    print(f"# one step: {len(step)} launches, span {(t1 - t0) / 1e6:.3f} ms, kernel time {busy / 1e3:.3f} ms")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"{k[0]:58s} wg {k[1]:6d} n {v[0]:4d} {v[1]:9.1f} us  avg {v[1] / v[0]:7.1f}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
