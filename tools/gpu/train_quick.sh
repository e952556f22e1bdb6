# training-step work: operator / gradient parity tests, then the step timing (launch by launch and as one graph replay)
cd $GRAFT_REPO_ROOT
O=gpurun_out/train_quick; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_train_embedding_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log | cut -c1-220
timeout 300 python tools/train_bench.py 64 512 2>&1 | grep "B="
