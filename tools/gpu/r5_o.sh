cd $GRAFT_REPO_ROOT
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())"
for pr in "" 1 0 -1; do
for v in "1 2" "0 0"; do
  set -- $v
  if [ -z "$pr" ]; then unset MKWS_TRAIN_SIDE_PRIORITY; else export MKWS_TRAIN_SIDE_PRIORITY=$pr; fi
  echo "== side priority '${pr:-default}' MKWS_TRAIN_GEMM2=$1 MKWS_TRAIN_GEMM_TN2=$2"; MKWS_TRAIN_BENCH_NO_GRAPH=1 MKWS_TRAIN_GEMM2=$1 MKWS_TRAIN_GEMM_TN2=$2 timeout 300 python tools/train_bench.py 64 512 2>&1 | grep "B=" | cut -c1-70
done
done
