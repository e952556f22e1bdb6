cd $GRAFT_REPO_ROOT
for L in "" multilingual_kws_amd/lib/libmkws_hip_trainhead.so; do
  if [ -n "$L" ]; then export MKWS_LIB=$PWD/$L; fi
  echo "== lib: ${L:-default}"
  for seed in 11 12 13 14 15; do
  timeout 300 python tools/train_graph_diff.py 3 $seed 2>&1 | grep -E "last-step|params:" | tr '\n' ' '; echo
  done
done
