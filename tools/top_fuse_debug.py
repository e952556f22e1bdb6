import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multilingual_kws_amd import weights
from multilingual_kws_amd.embedding_model import EmbeddingModel
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
em = EmbeddingModel(weights.synthetic_blob(), max_batch=1024)
rng = np.random.default_rng(21)
x = torch.from_numpy(rng.integers(0, 670, size=(B, 49, 40)).astype(np.float32) * np.float32(10 / 256)).cuda()
a = em.tap(x, "gap").clone().reshape(B, 1280)
em.set_option("fuse_top", 0)
b = em.tap(x, "gap").clone().reshape(B, 1280)
d = (a - b).abs().cpu().numpy()
print("max diff", d.max(), "rel", d.max() / float(b.abs().max()), "n differing", int((d > 0).sum()), "of", d.size)
cols = np.nonzero((d > 0).any(0))[0]; rows = np.nonzero((d > 0).any(1))[0]
print("rows differing", rows[:20], "cols differing count", len(cols), cols[:40])
print("col % 16 hist", np.bincount(cols % 16, minlength=16), "col//16 %8 hist", np.bincount((cols // 16) % 8, minlength=8))
i, j = np.unravel_index(d.argmax(), d.shape)
print("worst", i, j, float(a[i, j]), float(b[i, j]))
