"""Step time of the `backprop_into_embedding=True` phase (training-mode forward + backward + Adam of the whole embedding
and the head) on synthetic spectrogram-like inputs; not a BASELINE config (bench.py covers those)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from multilingual_kws_amd import weights
from multilingual_kws_amd.embedding_trainer import EmbeddingTrainer, TrainStepGraph
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
    n = 10
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    t1 = time.perf_counter()
    for _ in range(n):
        tr.forward_train(spec)
    torch.cuda.synchronize()
    df = (time.perf_counter() - t1) / n
    print(f"B={B}: launch by launch {dt * 1e3:.2f} ms/step = {B / dt:.0f} clips/s (training-mode forward alone {df * 1e3:.2f} ms); loss {hd.loss_grad(tr.forward_train(spec), labels).tolist()[0] / B:.4f}")
    if os.environ.get("MKWS_TRAIN_BENCH_NO_GRAPH"):
        continue
    for mode, what in (("tape", "recorded call tape, two streams"), ("hipgraph", "one hipGraph replay per step, one stream")):
        g = TrainStepGraph(tr, hd, B, 1e-4, mode=mode)
        for _ in range(3):
            g.run(spec, labels)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            g.run(spec, labels)
        torch.cuda.synchronize()
        dg = (time.perf_counter() - t0) / n
        print(f"B={B}: {what} {dg * 1e3:.2f} ms/step = {B / dg:.0f} clips/s ({len(g._tape) if g._tape else 0} library calls); loss {g.stats.tolist()[0] / B:.4f}", flush=True)
