#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 300 python scripts/ndac_bench.py > $O/ndac_bench.json 2> $O/ndac_bench.err; cat $O/ndac_bench.json; tail -3 $O/ndac_bench.err
rm -rf $O/prof_ndac
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ndac -- python $GRAFT_REPO_ROOT/scripts/ndac_bench.py --iters 2 < /dev/null > $O/prof_ndac.log 2>&1); echo "stats rc=$?"
find $O/prof_ndac -name '*kernel_trace.csv' -size +20M -delete
python - <<'PY'
import csv, glob, os, collections
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out")
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(O + "/prof_ndac/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-60:]
        agg[k][0] += 1; agg[k][1] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"{k:62s} {n:6d} launches {us / 1e3:10.2f} ms total {us / n:9.1f} us avg")
PY
timeout 600 python -m pytest tests/test_hip_configs.py -m gpu -q -k "cfg5" < /dev/null 2>&1 | tail -5
