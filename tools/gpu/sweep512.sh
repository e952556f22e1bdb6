# tile sweep of the dense layers of a 512-clip handle (fine-tune config) through MKWS_GEMM_FORCE="Mmax,MT,NT,splitK"
cd $GRAFT_REPO_ROOT
for cfg in "0,1,2,1" "512,1,2,1" "512,2,2,1" "512,1,4,1" "512,2,4,1" "512,2,4,2" "512,2,2,2" "512,1,4,2" "512,1,2,2" "512,2,4,4"; do
  MKWS_GEMM_FORCE=$cfg timeout 200 python bench.py --no-cpu-baseline --steps 20 --config finetune 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('$cfg', d['ms_per_step'], {n: x['ms_per_step'] for n,x in k.items() if 'pw_gemm' in n or 'splitk' in n})"
done
