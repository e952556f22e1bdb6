"""Four identical passes of the hot path (frontend + embedding forward, B = 1024) -- the workload the PMC and
phase-timing runs wrap (tools/pmc_traffic.sh, tools/gpu_round.sh, MKWS_LIB=...timing.so)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multilingual_kws_amd import synth, weights
from multilingual_kws_amd.embedding_model import EmbeddingModel
from multilingual_kws_amd.frontend import Frontend

dev = torch.device("cuda:0")
B = int(os.environ.get("ONE_FWD_B", "1024"))      # 512 / 256: the handles of the fine-tune / streaming configs (other workgroup shapes)
em = EmbeddingModel(weights.synthetic_blob(), max_batch=B)
fe = Frontend(max_samples=16000)
audio = torch.from_numpy(synth.clips_float32(B)).to(dev)
for _ in range(4):
    em.forward(fe.forward(audio))
torch.cuda.synchronize()
