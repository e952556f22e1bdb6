"""Experiment: start the first-round workgroups of mbconv_front_kernel's second / third / fourth slot of every CU late, so that the
workgroups sharing a CU stop running their MFMA and their LDS / latency phases in lockstep.  Prints the per-kernel table per setting."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multilingual_kws_amd import weights
from multilingual_kws_amd.embedding_model import EmbeddingModel

dev = torch.device("cuda:0")
B = 1024
em = EmbeddingModel(weights.synthetic_blob(), max_batch=B)
x = torch.rand((B, 49, 40), device=dev) * 26
for st in [int(v) for v in (sys.argv[1:] or ["0", "100", "200", "300", "400", "0"])]:
    em.set_option("front_stagger", st)
    prof = em.profile(x, reps=20)
    tot = sum(v for _, _, v in prof)
    fr = [(k, v) for _, k, v in prof if "front" in k]
    print(f"stagger {st * 10} ns: forward {tot * 1e3:.1f} us; " + "; ".join(f"{k.split('<')[1][:12]} {v * 1e3:.1f}" for k, v in fr), flush=True)
