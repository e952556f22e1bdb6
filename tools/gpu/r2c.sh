set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2d
timeout 600 python -m pytest tests/test_embedding_gpu.py -x -q > gpurun_out/r2d/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2d/pytest.log
tail -15 gpurun_out/r2d/pytest.log
cat > /tmp/one.py <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from multilingual_kws_amd import synth, weights
from multilingual_kws_amd.embedding_model import EmbeddingModel
from multilingual_kws_amd.frontend import Frontend
B = 1024
em = EmbeddingModel(weights.synthetic_blob(), max_batch=B)
em.set_option("fuse_mid", int(sys.argv[1]))
fe = Frontend(max_samples=16000)
audio = torch.from_numpy(synth.clips_float32(B)).cuda()
for _ in range(3):
    em.forward(fe.forward(audio))
torch.cuda.synchronize()
PY
for v in 1 2; do
MKWS_LIB=$PWD/multilingual_kws_amd/lib/libmkws_hip_timing.so timeout 300 python /tmp/one.py $v > gpurun_out/r2d/timing$v.log 2>&1
grep -E "mid-timing" gpurun_out/r2d/timing$v.log | tail -5
done
for v in 1 2 0; do
timeout 300 python bench.py --no-cpu-baseline --steps 30 --opt fuse_mid=$v > gpurun_out/r2d/bench_mid$v.json 2> gpurun_out/r2d/bench_mid$v.err
done
python - <<'PY'
import json
for n in ("mid1","mid2","mid0"):
    try:
        d=json.load(open(f"gpurun_out/r2d/bench_{n}.json"))
        print(n, d["value"], d["ms_per_step"])
        for k,v in d["kernels"].items():
            if "mid" in k: print("   ", k, v)
    except Exception as e: print(n, "failed", e)
PY
