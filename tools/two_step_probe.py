"""Whole steps (frontend + embedding forward, 1024 clips) alternating between two streams / two handles: do the tails of one step's launches fill
with the other step's workgroups?   python tools/two_step_probe.py  -- ms per step, serial vs two steps in flight."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multilingual_kws_amd import synth, weights
from multilingual_kws_amd.embedding_model import EmbeddingModel
from multilingual_kws_amd.frontend import Frontend
B = 1024
dev = torch.device("cuda:0")
blob = weights.synthetic_blob()
ems = [EmbeddingModel(blob, max_batch=B) for _ in range(2)]
fe = Frontend(max_samples=16000)
audio = torch.from_numpy(synth.clips_float32(B)).to(dev)
specs = [torch.empty((B, 49, 40), device=dev) for _ in range(2)]
embs = [torch.empty((B, 1024), device=dev) for _ in range(2)]
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
def serial(n):
    for i in range(n):
        fe.forward(audio, out=specs[0]); ems[0].forward(specs[0], out=embs[0])
def two(n):
    for i in range(n):
        k = i & 1
        with torch.cuda.stream(streams[k]):
            fe.forward(audio, out=specs[k]); ems[k].forward(specs[k], out=embs[k])
def timeit(fn, n=300):
    fn(20); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(n); torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for rep in range(3):
    print(f"serial {timeit(serial):.4f} ms/step   two steps in flight {timeit(two):.4f} ms/step", flush=True)
torch.cuda.synchronize(); print("same result:", torch.equal(embs[0], embs[1]), "degraded:", [e.get_option("pair_degraded") for e in ems], [e.get_option("exchange_error") for e in ems])
