cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in 0 512 5 10; do
  MKWS_MID2B=$v timeout 300 python tools/kernel_table.py 1024 20 "block2b" 2>&1 | grep -A1 "pass 1" | grep block2b | awk -v v=$v '{print "MID2B="v, $0}'
done
done
MKWS_MID2B=5 timeout 300 python -m pytest tests/test_embedding_gpu.py -m gpu -q -x -k "every_stage" 2>&1 | tail -1
