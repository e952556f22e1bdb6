cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/r6_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r6_pytest_gpu.log
