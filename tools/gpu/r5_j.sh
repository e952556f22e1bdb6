cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_j; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest -m gpu -q -x tests/test_train_gpu.py > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for v in "1 2" "1 0"; do
  set -- $v
  MKWS_TRAIN_GEMM2=$1 MKWS_TRAIN_GEMM_TN2=$2 timeout 300 python tools/gemm_shapes.py 512 > $O/shapes512_g$1_t$2.txt 2>&1
  MKWS_TRAIN_GEMM2=$1 MKWS_TRAIN_GEMM_TN2=$2 timeout 300 python tools/gemm_shapes.py 64 > $O/shapes64_g$1_t$2.txt 2>&1
  echo "gemm2=$1 tn2=$2: 512: $(tail -1 $O/shapes512_g$1_t$2.txt)   64: $(tail -1 $O/shapes64_g$1_t$2.txt)"
done
paste <(cut -c1-30,58-75 $O/shapes512_g1_t2.txt) <(cut -c58-75 $O/shapes512_g1_t0.txt) | head -40
paste <(cut -c1-30,58-75 $O/shapes64_g1_t2.txt) <(cut -c58-75 $O/shapes64_g1_t0.txt) | head -40
