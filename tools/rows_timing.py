"""Phase profile of mbconv_rows_kernel (timing build: MKWS_LIB=.../libmkws_hip_timing.so [MKWS_ABLATE=mask] python tools/rows_timing.py [B]).
Runs the taps that end behind blocks 2b and 3b a few times; the library prints one [rows-timing] line per launch."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multilingual_kws_amd import weights
from multilingual_kws_amd.embedding_model import EmbeddingModel

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
em = EmbeddingModel(weights.synthetic_blob(), max_batch=B)
em.set_option("fuse_rows", int(os.environ.get("FUSE_ROWS", "1")))
x = torch.rand((B, 49, 40), device=torch.device("cuda:0")) * 26
for _ in range(3):
    em.tap(x, "block3b")
torch.cuda.synchronize()
