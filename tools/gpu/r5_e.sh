# round 5, call E: (1) gate folded into the chain's projection operand load vs the gate pass (library A/B, same call), bit-identity tests;
# (2) one AddressSanitizer pass (xnack+ build of mkws_embed.hip) over the embedding tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_e; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest -m gpu -q -x tests/test_embedding_gpu.py > $O/pytest_embed.log 2>&1; echo "pytest embedding rc=$?"; tail -3 $O/pytest_embed.log
KFILTER=chain bash tools/gpu/ablibs.sh - multilingual_kws_amd/lib/libmkws_hip_gatepass.so - multilingual_kws_amd/lib/libmkws_hip_gatepass.so
unset MKWS_LIB
ASAN_RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
echo "asan runtime: $ASAN_RT"
( export HSA_XNACK=1 LD_PRELOAD=$ASAN_RT ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0 MKWS_LIB=$GRAFT_REPO_ROOT/multilingual_kws_amd/lib/libmkws_hip_asan.so LD_LIBRARY_PATH=$(dirname $ASAN_RT):/opt/rocm/lib:$LD_LIBRARY_PATH
  timeout 900 python -m pytest -m gpu -q -x tests/test_embedding_gpu.py tests/test_guard_bands_gpu.py > $O/pytest_asan.log 2>&1; echo "pytest under ASAN rc=$?" )
tail -25 $O/pytest_asan.log | cut -c1-300
