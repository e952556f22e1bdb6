# 2a front kernel: one workgroup walks the clip's channel blocks (fuse_walk) -- parity + same-process A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_walk; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_embedding_gpu.py -x -q -m gpu -k "options_agree or every_stage or full_batch" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
AB_OPTION=fuse_walk timeout 300 python tools/kernel_table.py 1024 20 block2a > $O/table.txt 2>&1; grep -E "pass|block2a" $O/table.txt | tail -12
