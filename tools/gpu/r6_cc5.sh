cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_cc5; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_embedding_gpu.py -m gpu -x -q -k "cluster_chain or cluster_kernel or serving_handle" 2>&1 | tail -4 | tee $O/pytest.txt
for rep in 1 2; do
for l in new old; do
  if [ $l = old ]; then export MKWS_LIB=$PWD/multilingual_kws_amd/lib/libmkws_hip_flagx1.so; else unset MKWS_LIB; fi
  timeout 300 python tools/latency_ab.py fuse_cluster_chain 1 2 2>&1 | tail -4 | sed "s/^/$l: /" | tee -a $O/ab.txt
done
done
unset MKWS_LIB
for n in 2 4 6; do timeout 120 python tools/chain_concurrency_probe.py $n 200 2>&1 | grep -v "amdgpu.ids" | tail -7 | tee -a $O/conc.txt; done
timeout 120 python tools/chain_concurrency_probe.py 3 200 0 2>&1 | grep -v "amdgpu.ids" | tail -4 | tee -a $O/conc.txt
