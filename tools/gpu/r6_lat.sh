cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_lat; rm -rf $O; mkdir -p $O
for opt in fuse_mid fuse_back fuse_walk fuse_stem; do
  timeout 200 python tools/latency_ab.py $opt 1 2 2>&1 | grep round | tee -a $O/lat.txt
done
