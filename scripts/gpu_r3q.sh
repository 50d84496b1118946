#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_ndac.py -m gpu -q < /dev/null 2>&1 | tail -3
for b in 1 8; do
  timeout 300 python scripts/ndac_bench.py --batch $b > $O/ndac_bench_b$b.json 2> $O/ndac_bench_b$b.err
  python -c "
import json; r=json.load(open('$O/ndac_bench_b$b.json'))
print('B=$b', {k: (round(v['ms'],2), round(v['audio_seconds_per_second'])) for k,v in r.items() if isinstance(v, dict)}, r.get('encode_mfma_code_mismatch_fraction'), r.get('decode_vs_exact_rel_to_peak'))"
done
