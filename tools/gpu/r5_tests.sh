# full -m gpu suite
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r05_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r05_pytest_gpu.log
