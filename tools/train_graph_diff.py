"""Which tensors differ between three eager training steps and three hipGraph-replayed ones (tests/test_train_gpu.py::test_graph_replayed_...)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multilingual_kws_amd import weights
from multilingual_kws_amd.embedding_trainer import EmbeddingTrainer, TrainStepGraph, drop_connect_rates
from multilingual_kws_amd.head import Head
blob = weights.synthetic_blob()
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
B, lr = 8, 1e-4
specs = [torch.from_numpy(rng.integers(0, 670, size=(B, 49, 40)).astype(np.float32) * np.float32(10 / 256)).cuda() for _ in range(3)]
labels = [torch.from_numpy(rng.integers(0, 3, B).astype(np.int32)).cuda() for _ in range(3)]
masks = [{n: rng.uniform(0, 1, B) >= r for n, r in drop_connect_rates().items()} for _ in range(3)]
nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
def eager():
    tr, hd = EmbeddingTrainer(blob), Head(max_batch=B, seed=3)
    for x, y, mk in list(zip(specs, labels, masks))[:nsteps]:
        emb = tr.forward_train(x, mk); hd.loss_grad(emb, y); tr.backward(hd.input_grad(B)); g = tr.grads.clone(); hd.adam_step(lr=lr); tr.adam_step(lr=lr)
    return tr, g
tr_e, g_e = eager()
tr_e2, g_e2 = eager()
print("eager vs eager identical:", np.array_equal(tr_e.blob(), tr_e2.blob()), torch.equal(g_e, g_e2))
tr_g, hd_g = EmbeddingTrainer(blob), Head(max_batch=B, seed=3)
step = TrainStepGraph(tr_g, hd_g, B, lr)
for x, y, mk in list(zip(specs, labels, masks))[:nsteps]:
    step.run(x, y, mk)
torch.cuda.synchronize()
print("last-step gradients eager vs graph: max abs diff", float((g_e - tr_g.grads).abs().max()), "n differing", int((g_e != tr_g.grads).sum()))
ge, gg = g_e.cpu().numpy(), tr_g.grads.cpu().numpy()
rel = []
for name, t in tr_e.tensors.items():
    sl = slice(t["offset"], t["offset"] + t["count"])
    m = float(np.abs(ge[sl]).max())
    if m > 0: rel.append((float(np.abs(ge[sl] - gg[sl]).max()) / m, name, m))
print("largest per-tensor relative gradient differences (eager vs graph):")
for r, name, m in sorted(rel, reverse=True)[:12]:
    print(f"  {name:40s} {r:.3e}  (gmax {m:.3e})")
pe, pg = tr_e.blob(), tr_g.blob()
d = np.abs(pe - pg)
print("params: frac > 1e-6:", (d > 1e-6).mean(), "max", d.max())
rows = []
for name, t in tr_e.tensors.items():
    sl = slice(t["offset"], t["offset"] + t["count"])
    n = int((d[sl] > 1e-6).sum())
    if n: rows.append((n, name, t["count"]))
for n, name, c in sorted(rows, reverse=True)[:14]:
    print(f"  {name:40s} {n:8d} of {c}")
