cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_cc; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_embedding_gpu.py -m gpu -x -q -k "cluster_chain" 2>&1 | tail -15 | tee $O/pytest_chain.txt
timeout 900 python -m pytest tests/test_embedding_gpu.py -m gpu -x -q -k "serving_handle or cluster_kernel or options_agree or every_stage" 2>&1 | tail -8 | tee $O/pytest_plans.txt
timeout 300 python tools/latency_ab.py fuse_cluster_chain 1 3 2>&1 | tail -6 | tee $O/ab_chain.txt
timeout 300 python tools/latency_ab.py fuse_back 1 2 2>&1 | tail -4 | tee $O/ab_back.txt
