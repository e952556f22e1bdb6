# gemv_kernel for live-serving handles: parity + batch-1 latency A/B (fuse_gemv 1 / 0 through an env knob of latency_ab)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_gemv; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_embedding_gpu.py tests/test_streaming.py -x -q -m gpu -k "serving_handle_plans or cluster or streaming or session or Session" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
timeout 300 python tools/latency_gemv_ab.py 2>&1 | tail -8
