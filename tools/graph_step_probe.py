"""Main config (frontend + embedding forward, 1024 clips): one step launch by launch vs one hipGraph replay per step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multilingual_kws_amd import synth, weights
from multilingual_kws_amd.embedding_model import EmbeddingModel
from multilingual_kws_amd.frontend import Frontend
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda:0")
em = EmbeddingModel(weights.synthetic_blob(), max_batch=B)
fe = Frontend(max_samples=16000)
audio = torch.from_numpy(synth.clips_float32(B)).to(dev)
spec = torch.empty((B, 49, 40), device=dev); emb = torch.empty((B, 1024), device=dev)
def step():
    fe.forward(audio, out=spec); em.forward(spec, out=emb)
for _ in range(20): step()
torch.cuda.synchronize()
def timeit(fn, n=300):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
side = torch.cuda.Stream(device=dev); side.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.stream(side): step()
torch.cuda.current_stream(dev).wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g): step()
ref = emb.clone(); g.replay(); torch.cuda.synchronize()
print("graph == eager:", torch.equal(ref, emb))
for rep in range(3):
    print(f"eager {timeit(step):.4f} ms/step   graph {timeit(g.replay):.4f} ms/step")
