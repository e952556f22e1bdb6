# tile sweep of the dense layers at M = 1024 through MKWS_GEMM_FORCE="Mmax,MT,NT,splitK" (all three dense layers take the forced tile)
cd $GRAFT_REPO_ROOT
for cfg in "0,1,2,1" "1024,2,4,1" "1024,2,2,1" "1024,1,4,1" "1024,2,4,2" "1024,1,2,1" "1024,2,2,2" "1024,1,4,2" "1024,2,3,1" "1024,2,6,1"; do
  MKWS_GEMM_FORCE=$cfg timeout 200 python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('$cfg', d['ms_per_step'], {n.replace('_kernel',''): x['ms_per_step'] for n,x in k.items() if 'pw_gemm' in n or 'splitk' in n})"
done
