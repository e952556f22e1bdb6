for cfg in "4096,2,4,1" "4096,2,4,2" "4096,2,4,4" "4096,2,2,1" "4096,2,2,2" "4096,2,2,4" "4096,1,4,2" "4096,1,2,2"; do
  echo "== $cfg"; MKWS_GEMM_FORCE=$cfg python tools/stage_prof.py 2>&1 | grep -E "^block6a  |^block6b  |^block7a  |^top|#reduce" | grep -E "block6a|block6b|block7a|top" | awk '{printf "%s %s %s | ", $1,$2,$3} END {print ""}'
done
for cfg in "12288,2,4,1" "12288,2,2,1" "12288,2,4,2" "12288,2,2,2" "12288,1,4,1" "12288,1,2,1"; do
  echo "== $cfg"; MKWS_GEMM_FORCE=$cfg python tools/stage_prof.py 2>&1 | grep -E "^block4a  |^block4b  |^block5a  |^block5b  |#reduce" | grep -E "block4a|block4b|block5a|block5b" | awk '{printf "%s %s %s | ", $1,$2,$3} END {print ""}'
done
