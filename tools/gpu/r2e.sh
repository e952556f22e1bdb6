set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2e
timeout 900 python -m pytest tests/test_train_gpu.py -x -q > gpurun_out/r2e/pytest_train.log 2>&1; echo "rc=$?" >> gpurun_out/r2e/pytest_train.log
tail -30 gpurun_out/r2e/pytest_train.log


