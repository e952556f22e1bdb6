# parity (embedding, streaming, training operators) + same-call A/B of all four configs against another build
#   gpurun -- 'bash tools/gpu/ab_all.sh multilingual_kws_amd/lib/libmkws_hip_prev.so'
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab_all; mkdir -p $O
timeout 900 python -m pytest tests/test_embedding_gpu.py tests/test_streaming.py tests/test_train_gpu.py tests/test_head_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
for c in embed finetune stream; do
  timeout 300 python bench.py --no-cpu-baseline --steps 50 --config $c > $O/${c}_new.json 2> $O/${c}_new.err
  MKWS_LIB=$PWD/$1 timeout 300 python bench.py --no-cpu-baseline --steps 50 --config $c > $O/${c}_prev.json 2> $O/${c}_prev.err
done
timeout 300 python bench.py --no-cpu-baseline --steps 50 > $O/embed_new2.json 2> $O/embed_new2.err
python - <<'PY'
import json, os
for c in ("embed", "finetune", "stream"):
    for v in ("new", "prev", "new2"):
        f=f"gpurun_out/ab_all/{c}_{v}.json"
        if not os.path.exists(f): continue
        try:
            d=json.load(open(f))
            ks={k.replace("mbconv_","").replace("_kernel",""): x["ms_per_step"] for k, x in d["kernels"].items()}
            print(c, v, d["value"], d["ms_per_step"], d.get("latency_ms_batch1"), {k: ks[k] for k in list(ks)[:18]} if c=="embed" else "")
        except Exception as e: print(c, v, "failed", e)
PY
