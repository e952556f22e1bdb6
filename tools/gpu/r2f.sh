set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2f
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r2f/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r2f/pytest.log
tail -6 gpurun_out/r2f/pytest.log
MKWS_LIB=$PWD/multilingual_kws_amd/lib/libmkws_hip_timing.so timeout 300 python tools/one_fwd.py > gpurun_out/r2f/timing.log 2>&1
grep -E "block-timing|mid-timing|front-timing" gpurun_out/r2f/timing.log | tail -24
timeout 300 python bench.py --no-cpu-baseline --steps 30 > gpurun_out/r2f/bench.json 2> gpurun_out/r2f/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r2f/bench.json"))
print(d["value"], d["ms_per_step"])
for k,v in d["kernels"].items(): print("   ", k, v)
PY
