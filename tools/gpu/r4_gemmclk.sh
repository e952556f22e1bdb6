cd $GRAFT_REPO_ROOT
MKWS_LIB=$GRAFT_REPO_ROOT/multilingual_kws_amd/lib/libmkws_hip_timing.so timeout 300 python tools/one_fwd.py 2>&1 | grep "gemm-timing\|block-timing\] shader" | tail -8
