cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2g
for v in 0 2; do
timeout 300 python bench.py --no-cpu-baseline --steps 30 --opt fuse_block=$v > gpurun_out/r2g/bench_fb$v.json 2> gpurun_out/r2g/bench_fb$v.err
done
python - <<'PY'
import json
for n in ("fb0","fb2"):
    d=json.load(open(f"gpurun_out/r2g/bench_{n}.json"))
    print(n, d["value"], d["ms_per_step"])
import subprocess
PY
python - <<'PY'
import os, sys, ctypes
sys.path.insert(0, os.getcwd())
import torch
from multilingual_kws_amd import synth, weights
from multilingual_kws_amd.embedding_model import EmbeddingModel
from multilingual_kws_amd.frontend import Frontend
B=1024
em = EmbeddingModel(weights.synthetic_blob(), max_batch=B)
em.set_option("fuse_block", 0)
fe = Frontend(max_samples=16000)
spec = fe.forward(torch.from_numpy(synth.clips_float32(B)).cuda())
for stage, kernel, ms in em.profile(spec, reps=5):
    if stage[5] in "4567" or stage.startswith("top"): print(f"{stage:22s} {kernel:40s} {ms*1000:8.1f} us")
PY
