cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_cc4; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_embedding_gpu.py -m gpu -x -q -k "cluster_chain or cluster_kernel or serving_handle" 2>&1 | tail -4 | tee $O/pytest.txt
MKWS_LIB=$PWD/multilingual_kws_amd/lib/libmkws_hip_timing.so timeout 300 python tools/one_fwd_small.py 1 2>&1 | grep -E "cluster-chain" | tail -10 | tee $O/chain_timing.txt
timeout 300 python tools/latency_ab.py fuse_cluster_chain 1 3 2>&1 | tail -6 | tee $O/ab_chain.txt
for n in 2 4; do timeout 120 python tools/chain_concurrency_probe.py $n 200 2>&1 | grep -v "amdgpu.ids" | tail -6 | tee -a $O/conc.txt; done
