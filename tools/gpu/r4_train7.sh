# single-launch BatchNorm backward for layers with few rows: tests, then A/B in one call
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_train7; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_hf_efficientnet_train_golden.py tests/test_train_embedding_gpu.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -2
for rep in 1 2; do
for F in 0 1; do
echo "-- bn_small=$F"; MKWS_TRAIN_BENCH_NO_GRAPH=1 MKWS_TRAIN_BN_SMALL=$F timeout 300 python tools/train_bench.py 64 512 2>&1 | grep "B="
done; done
