# training step: all training tests, step times
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_train3; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_train_embedding_gpu.py tests/test_hf_efficientnet_train_golden.py tests/test_pipeline_gpu.py tests/test_multigpu.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -3 | cut -c1-300
timeout 300 python tools/train_bench.py 64 512 2>&1 | grep "B=" | tee $O/train_bench.txt
