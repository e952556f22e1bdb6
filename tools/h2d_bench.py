"""PCIe-inclusive rate of the hot path: the caller's clips start in pinned host memory.
  serial:      H2D copy, then frontend + embedding, same stream
  overlapped:  double-buffered -- batch i+1 is copied on a second stream while batch i computes"""
import os
import sys
import time
sys.path.insert(0, os.getcwd())
import torch
from multilingual_kws_amd import synth, weights
from multilingual_kws_amd.embedding_model import EmbeddingModel
from multilingual_kws_amd.frontend import Frontend

B = 1024
dev = torch.device("cuda:0")
fe, em = Frontend(max_samples=16000), EmbeddingModel(weights.synthetic_blob(), max_batch=B)
host = torch.from_numpy(synth.clips_float32(B)).pin_memory()
d0, d1 = torch.empty_like(host, device=dev), torch.empty_like(host, device=dev)
out = torch.empty((B, 1024), device=dev)
N = 50
for _ in range(5):
    d0.copy_(host, non_blocking=True); em.forward(fe.forward(d0), out=out)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N):
    d0.copy_(host, non_blocking=True)
torch.cuda.synchronize(); tc = (time.perf_counter() - t0) / N
print(f"H2D of {host.numel() * 4 / 1e6:.1f} MB pinned: {tc * 1e3:.3f} ms = {host.numel() * 4 / tc / 1e9:.1f} GB/s")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N):
    d0.copy_(host, non_blocking=True); em.forward(fe.forward(d0), out=out)
torch.cuda.synchronize(); ts = (time.perf_counter() - t0) / N
print(f"serial copy + compute: {ts * 1e3:.3f} ms/step = {B / ts:.0f} clips/s")
copy_stream = torch.cuda.Stream()
bufs, evs = [d0, d1], [torch.cuda.Event(), torch.cuda.Event()]
done = [torch.cuda.Event(), torch.cuda.Event()]
with torch.cuda.stream(copy_stream):
    bufs[0].copy_(host, non_blocking=True); evs[0].record(copy_stream)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(N):
    cur, nxt = i & 1, (i + 1) & 1
    with torch.cuda.stream(copy_stream):
        copy_stream.wait_event(done[nxt]) if i >= 1 else None       # the buffer's previous consumer has finished
        bufs[nxt].copy_(host, non_blocking=True); evs[nxt].record(copy_stream)
    torch.cuda.current_stream().wait_event(evs[cur])
    em.forward(fe.forward(bufs[cur]), out=out)
    done[cur].record()
torch.cuda.synchronize(); to = (time.perf_counter() - t0) / N
print(f"double-buffered copy || compute: {to * 1e3:.3f} ms/step = {B / to:.0f} clips/s")
