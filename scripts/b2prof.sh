cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
for b in 2 8; do
rm -rf $O/prof_b$b
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_b$b -- python $GRAFT_REPO_ROOT/bench.py --batch $b --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-e2e < /dev/null > /dev/null 2>&1)
python - <<PY
import csv,glob,collections
f=glob.glob('$O/prof_b$b/**/*kernel_trace.csv',recursive=True)[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'conv_wino4' in r['Kernel_Name']:
        agg[(r['Kernel_Name'][:70], int(r['Grid_Size_X'])//512)].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1])):
    print('B=$b', k[0][-40:], 'WGs', k[1], 'n', len(v), 'avg us', round(sum(v)/len(v),1), 'us per tile-round', round(sum(v)/len(v)/ (k[1]/256),2))
PY
find $O/prof_b$b -name '*.csv' -delete
done
