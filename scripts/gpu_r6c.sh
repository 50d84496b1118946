#!/bin/bash
# round 6, call C: (1) cache-policy variants of the F(4,3) kernel (non-temporal hint on the streamed activations: halo / residual / shortcut
# loads, output stores; weights keep the default policy) -- parity per variant, then the POWER-AWARE interleaved cfg 2 A/B;
# (2) enhance_cli on a 64-file corpus with and without ragged batching.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
for v in nth nto ntho ntall; do
  echo "== parity $v"; FLOWDEC_HIP_LIB=$GRAFT_REPO_ROOT/flowdec_amd/variants/libflowdec_$v.so timeout 300 python scripts/wino4_check.py --no-timing 2>&1 | grep -c OK
done 2>&1 | tee $O/r6c_nt_parity.txt
bash scripts/ab_bench_libs.sh 3 hip nth nto ntho ntall 2>&1 | tee $O/r6c_ab_nt.txt
timeout 900 python scripts/cli_corpus_rtf.py --out $O/r06_cli_corpus_rtf.json 2>&1 | grep -v "rtf =" | tail -8
