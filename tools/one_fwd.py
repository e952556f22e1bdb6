import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from multilingual_kws_amd import weights
from multilingual_kws_amd.embedding_model import EmbeddingModel
dev = torch.device("cuda:0")
blob = weights.synthetic_blob()
B = 1024
em = EmbeddingModel(blob, max_batch=B)
rng = np.random.default_rng(0)
x = torch.from_numpy((rng.integers(0, 670, size=(B, 49, 40)).astype(np.float32) * np.float32(10/256))).to(dev)
for _ in range(3):
    em.forward(x)
torch.cuda.synchronize()
