"""Is the two-stream training step bound by the host or by the device?  Per batch size: host time to ENQUEUE a step (no synchronisation inside
the timed loop) against the synchronised step time, launch by launch and as the recorded call tape; and the single-stream hipGraph replay
(pure device time of the serialized launches).   python tools/train_host_probe.py [B ...]"""
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

    def timed(fn, n=10):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        host = (time.perf_counter() - t0) / n
        torch.cuda.synchronize()
        return host * 1e3, (time.perf_counter() - t0) / n * 1e3
    h, t = timed(step)
    print(f"B={B}: launch by launch: host enqueue {h:.2f} ms/step, synchronised {t:.2f} ms/step", flush=True)
    for mode in ("tape", "hipgraph"):
        g = TrainStepGraph(tr, hd, B, 1e-4, mode=mode)
        h, t = timed(lambda: g.run(spec, labels))
        print(f"B={B}: {mode}: host enqueue {h:.2f} ms/step, synchronised {t:.2f} ms/step", flush=True)
    tr.overlap_wgrad = False
    h, t = timed(step)
    print(f"B={B}: launch by launch, ONE stream: host enqueue {h:.2f} ms/step, synchronised {t:.2f} ms/step", flush=True)
