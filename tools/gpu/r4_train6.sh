# BatchNorm statistics inside the producing launch: A/B in one call (0 none, 1 GEMM epilogue, 3 + 3x3 depthwise, 7 + 5x5 depthwise), twice
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for F in 0 1 3 7; do
echo "-- fuse_bn_stats=$F"; MKWS_TRAIN_BENCH_NO_GRAPH=1 MKWS_TRAIN_FUSE_BN_STATS=$F timeout 300 python tools/train_bench.py 64 512 2>&1 | grep "B="
done; done
