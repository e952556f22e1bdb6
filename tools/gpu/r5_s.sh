cd $GRAFT_REPO_ROOT
for f in "" "32,1,2,8" "32,1,1,8" "32,1,1,16" "32,1,2,16" "32,1,4,8" ""; do
  if [ -z "$f" ]; then unset MKWS_GEMM_FORCE; else export MKWS_GEMM_FORCE=$f; fi
  timeout 200 python tools/latency_gemv_probe.py 2>&1 | grep "per window" | tail -2
done
