"""Host and device time of the grouped fine-tune step (BASELINE configs[3], transfer_learning.FrozenHeadTrainer): host enqueue time of the draws /
the assembly launches / the embedding forward / the optimizer steps of one group WITHOUT synchronising (the GPU runs behind), the synchronised
group time, and hipEvent times of the device work of each part.   python tools/finetune_group_profile.py [batch] [group]"""
import os
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

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
G = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
em = EmbeddingModel(weights.synthetic_blob(), max_batch=B * G)
emb = torch.empty((B * G, 1024), device=dev)
d = synth.write_fewshot_dataset(tempfile.mkdtemp(prefix="mkws_ft_"))
ms = input_data.standard_microspeech_model_settings(3)
ds = input_data.AudioDataset(ms, ["target"], d["bg_dir"], d["unknown"], unknown_percentage=50.0, spec_aug_params=input_data.SpecAugParams(percentage=80), seed=1)
tds = ds.init_single_target(input_data.AUTOTUNE, d["train"], is_training=True).shuffle(1000).repeat().batch(B)
groups = input_data.BatchGroups(tds)
head = Head(params=np.random.default_rng(0).uniform(-0.07, 0.07, 18507).astype(np.float32), max_batch=B, device=dev)


def group():
    t0 = time.perf_counter()
    drawn = [ds._draw_batch(tds, groups._next_indices(), []) for _ in range(G)]
    t1 = time.perf_counter()
    spec, labels = ds._assemble(tds, drawn)
    labels = labels.to(torch.int32)
    t2 = time.perf_counter()
    e = em.forward(spec, out=emb)
    t3 = time.perf_counter()
    for j in range(G):
        parallel.dp_step(head, e[j * B:(j + 1) * B], labels[j * B:(j + 1) * B], lr=1e-3)
    return t1 - t0, t2 - t1, t3 - t2, time.perf_counter() - t3


for _ in range(30):
    group()
torch.cuda.synchronize()
N = 200
acc = np.zeros(4)
t0 = time.perf_counter()
for _ in range(N):
    acc += group()
host = time.perf_counter() - t0
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print(f"B={B} G={G}: {tot / N / G * 1e3:.4f} ms per optimizer step ({B * G * N / tot:.0f} clips/s); host enqueue per group {host / N * 1e3:.3f} ms = draws "
      f"{acc[0] / N * 1e3:.3f} + assembly launches {acc[1] / N * 1e3:.3f} + embedding launches {acc[2] / N * 1e3:.3f} + {G} optimizer steps {acc[3] / N * 1e3:.3f}")


def ev(fn, reps=50):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


drawn = [ds._draw_batch(tds, groups._next_indices(), []) for _ in range(G)]
spec, labels = ds._assemble(tds, drawn)
l32 = labels.to(torch.int32)
e = em.forward(spec, out=emb)
print(f"device, per group: assembly (copy + augmentation + frontend + SpecAugment, {B * G} clips) {ev(lambda: ds._assemble(tds, drawn)):.4f} ms; "
      f"embedding {ev(lambda: em.forward(spec, out=emb)):.4f} ms; one optimizer step (loss/gradient + Adam, {B} rows) "
      f"{ev(lambda: parallel.dp_step(head, e[:B], l32[:B], lr=1e-3)):.4f} ms; loss/gradient only {ev(lambda: head.loss_grad(e[:B], l32[:B])):.4f} ms")
if os.environ.get("MKWS_FT_CPROFILE", "1") != "0":
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(100):
        drawn = [ds._draw_batch(tds, groups._next_indices(), []) for _ in range(G)]
        ds._assemble(tds, drawn)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(16)
