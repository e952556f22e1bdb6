"""Per-kernel table of one embedding forward (hipEvent pairs around every launch, mkws_embed_profile): python tools/kernel_table.py [B] [reps] [filter]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multilingual_kws_amd import weights
from multilingual_kws_amd.embedding_model import EmbeddingModel

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
flt = sys.argv[3] if len(sys.argv) > 3 else ""
em = EmbeddingModel(weights.synthetic_blob(), max_batch=B)
x = torch.rand((B, 49, 40), device=torch.device("cuda:0")) * 26
em.profile(x, reps=3)
OPT = os.environ.get("AB_OPTION")          # AB_OPTION=name: alternate the option between 0 and 1 (or AB_VALUES=a,b)
VALS = [int(v) for v in os.environ.get("AB_VALUES", "0,1").split(",")]
for rnd in range(4 if OPT else 2):
    if OPT:
        em.set_option(OPT, VALS[rnd & 1])
    prof = em.profile(x, reps=reps)
    print(f"pass {rnd}{(' ' + OPT + '=' + str(VALS[rnd & 1])) if OPT else ''}: forward {sum(v for _, _, v in prof) * 1e3:.1f} us")
    for stage, kernel, ms in prof:
        if flt in kernel or flt in stage:
            print(f"  {stage:14s} {kernel:48s} {ms * 1e3:7.1f} us")
