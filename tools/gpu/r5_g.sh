# round 5, call G: head training kernels with more loads in flight (library A/B) + the head tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_g; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest -m gpu -q tests/test_head_gpu.py tests/test_finetune_gpu.py > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for rep in 1 2; do
for l in "" multilingual_kws_amd/lib/libmkws_hip_oldhead.so; do
  if [ -z "$l" ]; then unset MKWS_LIB; else export MKWS_LIB=$PWD/$l; fi
  echo "== lib: ${l:-shipped}"
  MKWS_FT_CPROFILE=0 timeout 300 python tools/finetune_group_profile.py 2>&1 | grep "B=512\|device"
done
done
unset MKWS_LIB
