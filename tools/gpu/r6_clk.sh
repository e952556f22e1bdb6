cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_clk; rm -rf $O; mkdir -p $O
MKWS_LIB=$PWD/multilingual_kws_amd/lib/libmkws_hip_timing.so timeout 300 python tools/one_fwd_small.py 1 fuse_cluster_chain=0 2>&1 | grep -E "cluster-timing" | tail -10 | tee $O/clk.txt
