cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for v in NOMFMA NOPROD; do echo "--- $v"; FLOWDEC_HIP_LIB=$GRAFT_REPO_ROOT/flowdec_amd/variants/libflowdec_$v.so timeout 600 python scripts/wino44f_timing2.py 2>&1 | grep "F(4x4)"; done
