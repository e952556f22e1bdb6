cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_pairs; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_embedding_gpu.py -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest.txt
timeout 600 python tools/kernel_table.py 1024 20 chain 2>&1 | grep -E "chain|forward" | tee $O/t1024.txt
timeout 600 python tools/kernel_table.py 512 20 chain 2>&1 | grep -E "chain|forward" | tee $O/t512.txt
