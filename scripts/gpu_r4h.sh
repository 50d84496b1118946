#!/bin/bash
# round 4, call H: the F(4,3) kernel's evidence from the final code -- parity + per-launch A/B against the direct and F(2,3) kernels on every
# layer shape, SQ / TA / TCP / TCC counters on two shapes, per-workgroup phase split and the timing-only ablation variants
# (variants built locally: scripts/build_t2.sh, scripts/build_w4var.sh after `patch -p0 < scripts/conv_wino4_ablation.patch`)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 400 python scripts/wino4_check.py --with-wino2 2>&1 | grep -v amdgpu.ids > $O/r4h_wino4_vs_direct.txt; tail -12 $O/r4h_wino4_vs_direct.txt
for S in 1 2; do SHAPE=$S bash scripts/pmc_wino4.sh 2>&1 | grep -v "^rc=" > $O/r4h_pmc_shape$S.txt; grep -- "->" $O/r4h_pmc_shape$S.txt; done
SHAPE=1 bash scripts/pmc_wino4_mem.sh 2>&1 | grep -v "^set" > $O/r4h_pmc_mem_shape1.txt; tail -30 $O/r4h_pmc_mem_shape1.txt
for v in t2 nowdma nohalo nobar noprod notrans noconv ring3 mfmaonly; do
  echo "== variant $v"
  FLOWDEC_HIP_LIB=$GRAFT_REPO_ROOT/flowdec_amd/variants/libflowdec_$v.so timeout 200 python scripts/wino4_timing2.py < /dev/null 2>&1 | grep "wino4\|direct" | cut -c1-150
done > $O/r4h_variants.txt; cat $O/r4h_variants.txt
