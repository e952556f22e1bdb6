cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_se4; rm -rf $O; mkdir -p $O
timeout 100 tools/microbench/_bin/se4_unit | grep -E "max error|launch" | tee $O/unit.txt
timeout 300 python tools/se4_debug.py 5 2>&1 | grep -E "max" | tee $O/debug.txt
timeout 1500 python -m pytest tests/test_embedding_gpu.py -m gpu -x -q 2>&1 | tail -8 | tee $O/pytest.txt
AB_OPTION=fuse_se4 timeout 600 python tools/kernel_table.py 1024 > $O/ab1024.txt 2>&1; grep -E "chain|forward" $O/ab1024.txt | head -30
AB_OPTION=fuse_se4 timeout 600 python tools/kernel_table.py 256 > $O/ab256.txt 2>&1; grep -E "chain_kernel|forward" $O/ab256.txt | head
