cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "X=0" "MKWS_GEMM_KS2=2" "MKWS_GEMM_FORCE=1024,2,4,1"; do
  env $v timeout 300 python tools/kernel_table.py 1024 30 "gemm" 2>&1 | grep -A4 "pass 1" | grep -E "dense" | awk '{printf "%s %s %s | ", $1, $2, $3}' | sed "s/^/[$v] /"; echo
done
done
MKWS_GEMM_KS2=2 timeout 600 python -m pytest tests/test_embedding_gpu.py -m gpu -q -x -k "every_stage or ragged or full_batch" 2>&1 | tail -2
