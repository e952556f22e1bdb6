set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2b
timeout 600 python -m pytest tests/test_embedding_gpu.py -x -q > gpurun_out/r2b/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2b/pytest.log
tail -15 gpurun_out/r2b/pytest.log
MKWS_LIB=$PWD/multilingual_kws_amd/lib/libmkws_hip_timing.so timeout 300 python tools/one_fwd.py > gpurun_out/r2b/timing.log 2>&1
grep -E "mid-timing|front-timing" gpurun_out/r2b/timing.log | tail -12
timeout 300 python bench.py --no-cpu-baseline --steps 30 > gpurun_out/r2b/bench_mid1.json 2> gpurun_out/r2b/bench_mid1.err
timeout 300 python bench.py --no-cpu-baseline --steps 30 --opt fuse_mid=0 > gpurun_out/r2b/bench_mid0.json 2> gpurun_out/r2b/bench_mid0.err
python - <<'PY'
import json
for n in ("mid1","mid0"):
    try:
        d=json.load(open(f"gpurun_out/r2b/bench_{n}.json"))
        print(n, d["value"], d["ms_per_step"])
        for k,v in d["kernels"].items():
            if "mid" in k or "front" in k or "se_" in k or "true" in k: print("   ", k, v)
    except Exception as e: print(n, "failed", e)
PY
