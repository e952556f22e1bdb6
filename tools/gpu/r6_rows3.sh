# rows kernel iteration 3: both shapes of both blocks: parity, phase timing (timing build), per-kernel table
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_rows3; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_embedding_gpu.py -x -q -m gpu -k "register_resident" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
for fr in 1 2; do
  FUSE_ROWS=$fr MKWS_LIB=$PWD/multilingual_kws_amd/lib/libmkws_hip_timing.so timeout 300 python tools/rows_timing.py 1024 2>&1 | grep "rows-timing" | tail -2 >> $O/timing.txt
done
cat $O/timing.txt
AB_OPTION=fuse_rows AB_VALUES=1,2 timeout 300 python tools/kernel_table.py 1024 20 block > $O/table.txt 2>&1; grep -E "pass|block2b|block3b" $O/table.txt | tail -6
