"""Batch-1 embedding latency, graph replay with a sync per window, alternating an option between 0 and 1 in one process:
    python tools/latency_ab.py cluster_rotate [max_batch] [rounds]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multilingual_kws_amd import weights
from multilingual_kws_amd.embedding_model import EmbeddingModel

opt = sys.argv[1]
mb = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda:0")
blob = weights.synthetic_blob()
ems, graphs = {}, {}
x = torch.rand((mb, 49, 40), device=dev) * 26
out = torch.empty((mb, 1024), device=dev)
for v in (0, 1):
    em = ems[v] = EmbeddingModel(blob, max_batch=mb)
    em.set_option(opt, v)
    em.forward(x, out=out)
    torch.cuda.synchronize()
    g = graphs[v] = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        em.forward(x, out=out)
for r in range(rounds):
    for v in (0, 1):
        g = graphs[v]
        for _ in range(50):
            g.replay()
        torch.cuda.synchronize()
        n = 300
        t0 = time.perf_counter()
        for _ in range(n):
            g.replay()
            torch.cuda.synchronize()
        lat = (time.perf_counter() - t0) / n
        print(f"round {r} {opt}={v}: {lat * 1e3:.4f} ms per window (max_batch {mb})", flush=True)
