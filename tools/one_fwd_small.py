"""Two passes of the embedding forward on a small handle (phase timing of the cluster kernel: MKWS_LIB=...timing.so)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multilingual_kws_amd import weights
from multilingual_kws_amd.embedding_model import EmbeddingModel
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
em = EmbeddingModel(weights.synthetic_blob(), max_batch=B)
for kv in sys.argv[2:]:            # option=value pairs, e.g. fuse_cluster_chain=0
    k, v = kv.split("=")
    em.set_option(k, int(v))
x = torch.rand((B, 49, 40), device="cuda") * 26
for _ in range(3):
    em.forward(x)
torch.cuda.synchronize()
