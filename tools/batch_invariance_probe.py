import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from multilingual_kws_amd import weights, arch
from multilingual_kws_amd.embedding_model import EmbeddingModel
blob = weights.synthetic_blob()
rng = np.random.default_rng(21)
spec = (rng.integers(0, 670, size=(8, 49, 40)).astype(np.float32) * np.float32(10/256))
x = torch.from_numpy(spec).cuda()
em = EmbeddingModel(blob, max_batch=8)
names = []
for name, cin, cout, k, s, e in arch.BLOCKS:
    p = "block" + name
    names += ([p + "_expand"] if e != 1 else []) + [p + "_dw", p + "_gate", p]
names += ["top", "gap", "dense", "dense_1"]
for n in names:
    try:
        a = em.tap(x, n).cpu().numpy(); b = em.tap(x[:1], n).cpu().numpy()
    except Exception as ex:
        print(n, "ERR", str(ex)[:60]); continue
    per = a.size // 8
    same = np.array_equal(a[:per], b[:per])
    if not same:
        print(n, "DIFF max", np.abs(a[:per] - b[:per]).max())
        break
    else:
        print(n, "same")
