cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_se4; mkdir -p $O
timeout 1500 python -m pytest tests/test_embedding_gpu.py -m gpu -x -q -k every_stage 2>&1 | grep -v "^$" | head -80 | tee $O/pytest_fail.txt
