cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_train5; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_train_gpu.py -m gpu -q -x -k "one_operator or oracle or graph" > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -2
bash tools/gpu/r4_train4.sh 64
