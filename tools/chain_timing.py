"""Phase profile of the depth-fused chain against the single-block kernels (timing build: MKWS_LIB=.../libmkws_hip_timing.so).
python tools/chain_timing.py [B]   -- prints the [chain-timing] / [block-timing] lines of the LAST of three forwards per setting."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multilingual_kws_amd import weights
from multilingual_kws_amd.embedding_model import EmbeddingModel

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
em = EmbeddingModel(weights.synthetic_blob(), max_batch=B)
x = torch.rand((B, 49, 40), device=torch.device("cuda:0")) * 26
for chain in (1, 0, 1, 0):
    em.set_option("fuse_chain", chain)
    sys.stderr.write(f"==== fuse_chain={chain}\n"); sys.stderr.flush()
    for _ in range(3):
        sys.stderr.write("---- pass\n"); sys.stderr.flush()
        em.forward(x)
    torch.cuda.synchronize()
