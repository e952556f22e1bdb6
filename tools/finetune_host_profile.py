"""Where the fine-tune step's host time goes (BASELINE configs[3]): wall time of next(it) / em.forward / dp_step calls WITHOUT synchronising (= host
enqueue cost; the GPU runs behind), the synchronised step time, and a cProfile of 200 batch assemblies.   python tools/finetune_host_profile.py"""
import cProfile
import os
import pstats
import sys
import tempfile
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from multilingual_kws_amd import parallel, synth, weights
from multilingual_kws_amd.embedding import input_data
from multilingual_kws_amd.embedding_model import EmbeddingModel
from multilingual_kws_amd.head import Head

B = 512
dev = torch.device("cuda:0")
em = EmbeddingModel(weights.synthetic_blob(), max_batch=B)
emb = torch.empty((B, 1024), device=dev)
d = synth.write_fewshot_dataset(tempfile.mkdtemp(prefix="mkws_ft_"))
ms = input_data.standard_microspeech_model_settings(3)
ds = input_data.AudioDataset(ms, ["target"], d["bg_dir"], d["unknown"], unknown_percentage=50.0, spec_aug_params=input_data.SpecAugParams(percentage=80), seed=1)
it = iter(ds.init_single_target(input_data.AUTOTUNE, d["train"], is_training=True).shuffle(1000).repeat().batch(B))
head = Head(params=np.random.default_rng(0).uniform(-0.07, 0.07, 18507).astype(np.float32), max_batch=B, device=dev)
for _ in range(20):
    s, l = next(it); parallel.dp_step(head, em.forward(s, out=emb), l, lr=1e-3)
torch.cuda.synchronize()
N = 300
t = [0.0, 0.0, 0.0]
t0 = time.perf_counter()
for _ in range(N):
    a = time.perf_counter(); s, l = next(it)
    b = time.perf_counter(); e = em.forward(s, out=emb)
    c = time.perf_counter(); parallel.dp_step(head, e, l, lr=1e-3)
    dd = time.perf_counter()
    t[0] += b - a; t[1] += c - b; t[2] += dd - c
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print(f"async loop: {tot / N * 1e3:.3f} ms/step; host enqueue: next(it) {t[0] / N * 1e3:.3f}  em.forward {t[1] / N * 1e3:.3f}  dp_step {t[2] / N * 1e3:.3f} ms")
# GPU-only time of the same step (events around an already-assembled batch)
s, l = next(it)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(50):
    parallel.dp_step(head, em.forward(s, out=emb), l, lr=1e-3)
e1.record(); torch.cuda.synchronize()
print(f"embedding + head on a resident batch: {e0.elapsed_time(e1) / 50:.3f} ms")
pr = cProfile.Profile(); pr.enable()
for _ in range(200):
    next(it)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
