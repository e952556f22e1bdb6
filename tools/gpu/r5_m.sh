cd $GRAFT_REPO_ROOT
for v in "1 2" "0 0"; do
  set -- $v
  echo "== MKWS_TRAIN_GEMM2=$1 MKWS_TRAIN_GEMM_TN2=$2"; MKWS_TRAIN_GEMM2=$1 MKWS_TRAIN_GEMM_TN2=$2 timeout 300 python tools/train_host_probe.py 64 512 2>&1 | grep "B="
done
