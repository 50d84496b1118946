#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
rm -rf $O/prof_ndac; rm -rf gpurun_out/prof_ndac
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ndac -- python $GRAFT_REPO_ROOT/scripts/ndac_bench.py --iters 2 < /dev/null > $O/prof_ndac.log 2>&1); echo "stats rc=$?"
find $O/prof_ndac -name '*kernel_trace.csv' -size +20M -delete
python - <<'PY'
import csv, glob, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out")
for f in glob.glob(O + "/prof_ndac/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if float(r["Percentage"]) > 0.3:
            print(f"{r['Name'].replace('(anonymous namespace)::','').split('(')[0][:60]:62s} calls {int(r['Calls']):5d} total {float(r['TotalDurationNs'])/1e6:9.2f} ms avg {float(r['AverageNs'])/1e3:9.1f} us {float(r['Percentage']):6.2f} %")
PY
