cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_midpro; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_embedding_gpu.py -m gpu -x -q -k "every_stage or options_agree or full_batch or mid_batch" 2>&1 | tail -3 | tee $O/pytest.txt
for rep in 1 2 3; do
for l in new old; do
  if [ $l = old ]; then export MKWS_LIB=$PWD/multilingual_kws_amd/lib/libmkws_hip_old.so; else unset MKWS_LIB; fi
  timeout 300 python tools/kernel_table.py 1024 20 mid 2>&1 | grep -E "forward|mid" | tail -4 | sed "s/^/$l: /" | tee -a $O/t1024.txt
done
done
unset MKWS_LIB
