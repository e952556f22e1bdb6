import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from multilingual_kws_amd import weights, arch
from multilingual_kws_amd.embedding_model import EmbeddingModel
from oracle.efficientnet_oracle import EmbeddingOracle
dev = torch.device("cuda:0")
blob = weights.synthetic_blob()
B = 1024
em = EmbeddingModel(blob, max_batch=B)
rng = np.random.default_rng(0)
spec = (rng.integers(0, 670, size=(B, 49, 40)).astype(np.float32) * np.float32(10/256))
x = torch.from_numpy(spec).to(dev)
ref = EmbeddingOracle(blob).forward(spec[:4]).numpy()
got = em.forward(x)[:4].cpu().numpy()
print("rel err", np.abs(got-ref).max()/np.abs(ref).max())
em.profile(x, reps=2)
rows = em.profile(x, reps=10)
costs = arch.stage_costs(B)
tot = 0
for stage, kern, ms in rows:
    f, b = costs["stem_dw"] if kern == "stem_dw_kernel" else (0.0, 0.0) if stage.endswith("#reduce") else (costs[stage.replace("_dw", "_front")] if kern.startswith("mbconv_front") else (costs[stage + "_block"] if kern.startswith("mbconv_block") else costs[stage]))
    tot += ms
    print(f"{stage:16s} {kern:28s} {ms*1000:8.1f} us  {f/ms/1e9:7.1f} TF/s  {b/ms/1e6:7.0f} GB/s  ({f/1e9:.2f} GF, {b/1e6:.1f} MB)")
print("total ms", tot)
