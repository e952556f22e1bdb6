cd $GRAFT_REPO_ROOT
for L in "" multilingual_kws_amd/lib/libmkws_hip_nounroll.so; do
  if [ -n "$L" ]; then export MKWS_LIB=$PWD/$L; fi
  echo "== lib: ${L:-default}"
  timeout 600 python -m pytest tests/test_train_gpu.py -m gpu -q -k "graph_replayed" 2>&1 | grep -E "passed|failed|assert \(" | head -4
done
