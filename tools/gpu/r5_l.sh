cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_l; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest -m gpu -q -x tests/test_train_gpu.py tests/test_hf_efficientnet_train_golden.py tests/test_train_embedding_gpu.py > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
for v in "1 2" "0 0"; do
  set -- $v
  MKWS_TRAIN_GEMM2=$1 MKWS_TRAIN_GEMM_TN2=$2 timeout 300 python tools/gemm_shapes.py 512 > $O/shapes512_g$1_t$2.txt 2>&1
  MKWS_TRAIN_GEMM2=$1 MKWS_TRAIN_GEMM_TN2=$2 timeout 300 python tools/gemm_shapes.py 64 > $O/shapes64_g$1_t$2.txt 2>&1
  echo "gemm2=$1 tn2=$2: 512: $(tail -1 $O/shapes512_g$1_t$2.txt)   64: $(tail -1 $O/shapes64_g$1_t$2.txt)"
done


for v in "1 2" "0 0" "1 2" "0 0"; do
  set -- $v
  echo "== MKWS_TRAIN_GEMM2=$1 MKWS_TRAIN_GEMM_TN2=$2"; MKWS_TRAIN_GEMM2=$1 MKWS_TRAIN_GEMM_TN2=$2 timeout 300 python tools/train_bench.py 64 512 2>&1 | grep "B=" | grep "launch by launch"
done
for v in "1 2" "0 0"; do
  set -- $v
  echo "== host probe MKWS_TRAIN_GEMM2=$1 MKWS_TRAIN_GEMM_TN2=$2"; MKWS_TRAIN_GEMM2=$1 MKWS_TRAIN_GEMM_TN2=$2 timeout 300 python tools/train_host_probe.py 64 512 2>&1 | grep "B="
done
