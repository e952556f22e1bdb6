cd $GRAFT_REPO_ROOT
export MKWS_GEMM_FORCE_MIN=2048
for rep in 1 2; do
for v in "9,2,2,1" "4096,1,4,1" "4096,2,4,1" "4096,1,2,1" "4096,2,1,1" "4096,1,3,1" "4096,2,3,1"; do
  MKWS_GEMM_FORCE=$v timeout 300 python tools/kernel_table.py 1024 20 "top" 2>&1 | grep -A1 "pass 1" | grep top | awk -v v=$v '{print "FORCE="v, $0}'
done
done
