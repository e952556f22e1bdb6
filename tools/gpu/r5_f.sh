# round 5, call F: augmentation kernel with eight loads in flight per thread vs the plain loops (library A/B), tests that pin its values
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_f; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest -m gpu -q tests/test_pipeline_gpu.py tests/test_finetune_gpu.py > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for rep in 1 2; do
for l in "" multilingual_kws_amd/lib/libmkws_hip_oldaug.so; do
  if [ -z "$l" ]; then unset MKWS_LIB; else export MKWS_LIB=$PWD/$l; fi
  echo "== lib: ${l:-shipped}"
  MKWS_FT_CPROFILE=0 timeout 300 python tools/finetune_group_profile.py 2>&1 | grep "B=512\|device"
done
done
unset MKWS_LIB
