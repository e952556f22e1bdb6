# 512-clip handle: parity + same-call A/B of the fine-tune config against another build   gpurun -- 'bash tools/gpu/mid.sh [other.so]'
cd $GRAFT_REPO_ROOT
O=gpurun_out/mid; mkdir -p $O
timeout 900 python -m pytest tests/test_embedding_gpu.py tests/test_consumers.py tests/test_finetune_gpu.py -x -q -m gpu -k "mid_batch or graph or consumers or finetune_step" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
timeout 300 python bench.py --no-cpu-baseline --steps 30 --config finetune > $O/ft_new.json 2> $O/ft_new.err
if [ -n "$1" ]; then MKWS_LIB=$PWD/$1 timeout 300 python bench.py --no-cpu-baseline --steps 30 --config finetune > $O/ft_prev.json 2> $O/ft_prev.err; fi
python - <<'PY'
import json, os
for v in ("ft_new", "ft_prev"):
    f=f"gpurun_out/mid/{v}.json"
    if not os.path.exists(f): continue
    try:
        d=json.load(open(f))
        print(v, d["value"], d["ms_per_step"], {k.replace("mbconv_",""): x["ms_per_step"] for k, x in d["kernels"].items() if "pair" in k or "block_kernel" in k})
    except Exception as e: print(v, "failed", e)
PY
