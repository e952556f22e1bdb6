# training operators after the GEMM / fold work: full training test files + capi + gradient oracles, then the per-shape tables and the step (defaults)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_q; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest -m gpu -q tests/test_train_gpu.py tests/test_hf_efficientnet_train_golden.py tests/test_train_embedding_gpu.py tests/test_capi.py tests/test_multigpu.py > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 300 python tools/gemm_shapes.py 512 > $O/r05_gemm_shapes_512.txt 2>&1; tail -1 $O/r05_gemm_shapes_512.txt
timeout 300 python tools/gemm_shapes.py 64 > $O/r05_gemm_shapes_64.txt 2>&1; tail -1 $O/r05_gemm_shapes_64.txt
MKWS_TRAIN_GEMM_TN2=2 timeout 300 python tools/gemm_shapes.py 512 > $O/r05_gemm_shapes_512_tn_ring.txt 2>&1; tail -1 $O/r05_gemm_shapes_512_tn_ring.txt
MKWS_TRAIN_GEMM_TN2=2 timeout 300 python tools/gemm_shapes.py 64 > $O/r05_gemm_shapes_64_tn_ring.txt 2>&1; tail -1 $O/r05_gemm_shapes_64_tn_ring.txt
timeout 300 python tools/train_bench.py 64 512 2>&1 | grep "B=" > $O/r05_train_bench.txt; cat $O/r05_train_bench.txt
