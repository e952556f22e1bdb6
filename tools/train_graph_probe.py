"""Probe: how long does the backprop_into_embedding step take on the GPU once the host is out of the way?  Captures the
current step (fixed Adam step index: timing only, not a training run) in a hipGraph and replays it."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from multilingual_kws_amd import weights
from multilingual_kws_amd.embedding_trainer import EmbeddingTrainer
from multilingual_kws_amd.head import Head

for B in (int(a) for a in (sys.argv[1:] or ["64", "512"])):
    rng = np.random.default_rng(0)
    spec = torch.from_numpy(rng.integers(0, 670, size=(B, 49, 40)).astype(np.float32) * np.float32(10 / 256)).cuda()
    labels = torch.from_numpy(rng.integers(0, 3, B).astype(np.int32)).cuda()
    tr, hd = EmbeddingTrainer(weights.synthetic_blob()), Head(max_batch=B, seed=0)

    def step():
        emb = tr.forward_train(spec)
        hd.loss_grad(emb, labels)
        tr.backward(hd.input_grad(B))
        hd.adam_step(lr=1e-4)
        tr.adam_step(lr=1e-4)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 5
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"B={B}: eager {eager * 1e3:.2f} ms/step, graph replay {dt * 1e3:.2f} ms/step", flush=True)
