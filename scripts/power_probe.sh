#!/bin/bash
# samples clocks / power while the benchmark is running
cd $GRAFT_REPO_ROOT
(timeout 200 python bench.py --steps 60 --warmup 3 --no-cpu-baseline --no-roofline < /dev/null > /tmp/b.log 2>&1; echo done > /tmp/b.done) &
for i in $(seq 1 120); do
  [ -f /tmp/b.done ] && break
  rocm-smi --showpower --showclocks 2>/dev/null | grep -i "sclk\|Package Power" | awk '{printf "%s ", $NF; if (/Power/) printf "%s W ", $(NF)} END{print ""}' 
  sleep 0.5
done | sort | uniq -c | sort -rn | head -12
cat /tmp/b.log | tail -1 | cut -c1-200
