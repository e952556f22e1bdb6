# AddressSanitizer pass over the embedding kernels (library built by: hipcc --offload-arch=gfx950:xnack+ -fsanitize=address -shared-libsan ...)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_asan; rm -rf $O; mkdir -p $O
ASAN_RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
ls -la $ASAN_RT
rocminfo 2>/dev/null | grep -i "xnack\|gfx950" | head -5
for xn in 1 0; do
  env HSA_XNACK=$xn LD_PRELOAD=$ASAN_RT ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 MKWS_LIB=$GRAFT_REPO_ROOT/multilingual_kws_amd/lib/libmkws_hip_asan.so \
    timeout 600 python -X faulthandler tools/asan_pass.py > $O/out_xnack$xn.txt 2> $O/err_xnack$xn.txt
  echo "HSA_XNACK=$xn: rc=$?"; tail -12 $O/out_xnack$xn.txt; tail -15 $O/err_xnack$xn.txt | cut -c1-300
done
