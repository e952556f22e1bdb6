# round 5, call H: register-ring NN / NT training GEMM (train_gemm2_kernel) -- operator tests, gradient oracles, per-shape and whole-step A/B by MKWS_TRAIN_GEMM2
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_h; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest -m gpu -q -x tests/test_train_gpu.py tests/test_hf_efficientnet_train_golden.py tests/test_train_embedding_gpu.py > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
for v in 1 0; do
  MKWS_TRAIN_GEMM2=$v timeout 300 python tools/gemm_shapes.py 512 > $O/shapes512_g$v.txt 2>&1
  MKWS_TRAIN_GEMM2=$v timeout 300 python tools/gemm_shapes.py 64 > $O/shapes64_g$v.txt 2>&1
  echo "gemm2=$v: 512: $(tail -1 $O/shapes512_g$v.txt)   64: $(tail -1 $O/shapes64_g$v.txt)"
done
for v in 1 0 1 0; do
  echo "== MKWS_TRAIN_GEMM2=$v"; MKWS_TRAIN_GEMM2=$v timeout 300 python tools/train_bench.py 64 512 2>&1 | grep "B=" | grep "launch by launch\|call tape"
done
