# parity subset + same-call A/B of the shipped library against multilingual_kws_amd/lib/libmkws_hip_prev.so (KFILTER = kernel-name substring)
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab_prev; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_embedding_gpu.py -m gpu -q -x -k "${TESTS:-every_stage or options or ragged or full_batch or mid_batch or small_batch}" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
bash tools/gpu/ablibs.sh - multilingual_kws_amd/lib/libmkws_hip_prev.so - multilingual_kws_amd/lib/libmkws_hip_prev.so
