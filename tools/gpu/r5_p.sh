cd $GRAFT_REPO_ROOT
for v in "1 1" "1 0" "1 2" "1 1" "1 0" "0 0"; do
  set -- $v
  echo "== GEMM2=$1 TN2=$2"; MKWS_TRAIN_BENCH_NO_GRAPH=1 MKWS_TRAIN_GEMM2=$1 MKWS_TRAIN_GEMM_TN2=$2 timeout 300 python tools/train_bench.py 64 512 2>&1 | grep "B=" | cut -c1-70
done
