# whole-block / GEMM kernels: parity, same-call A/B against another build (arg 1, optional), per-phase timing
#   gpurun -- 'bash tools/gpu/pair.sh [multilingual_kws_amd/lib/libmkws_hip_prev.so]'
cd $GRAFT_REPO_ROOT
O=gpurun_out/pair; mkdir -p $O
timeout 900 python -m pytest tests/test_embedding_gpu.py tests/test_streaming.py -x -q -m gpu > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
timeout 300 python bench.py --no-cpu-baseline --steps 50 > $O/bench_new.json 2> $O/bench_new.err
if [ -n "$1" ]; then MKWS_LIB=$PWD/$1 timeout 300 python bench.py --no-cpu-baseline --steps 50 > $O/bench_prev.json 2> $O/bench_prev.err; fi
timeout 300 python bench.py --no-cpu-baseline --steps 50 > $O/bench_new2.json 2> $O/bench_new2.err
python - <<'PY'
import json, os
for v in ("bench_new", "bench_prev", "bench_new2"):
    f=f"gpurun_out/pair/{v}.json"
    if not os.path.exists(f): continue
    try:
        d=json.load(open(f))
        print(v, d["value"], d["ms_per_step"], {k.replace("mbconv_","").replace("_kernel",""): x["ms_per_step"] for k, x in d["kernels"].items() if "pair" in k or "block_kernel" in k or "pw_gemm" in k})
    except Exception as e: print(v, "failed", e)
PY
MKWS_LIB=$PWD/multilingual_kws_amd/lib/libmkws_hip_timing.so timeout 300 python tools/one_fwd.py 2>&1 | grep -E "block-timing\] block(4c|5b|6a|6c|7a)" | tail -5
