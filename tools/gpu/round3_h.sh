cd $GRAFT_REPO_ROOT
O=gpurun_out/r3h; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_embedding_gpu.py -m gpu -q -x -k "cluster or serving" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest.log | cut -c1-220
PLANS=whole+pair,cluster timeout 300 python tools/plan_sweep.py 1 4 16 32 64 2>&1 | grep -v amdgpu.ids
