import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from multilingual_kws_amd import weights
from oracle import efficientnet_oracle as eo
man = weights.manifest()
print("tensors", len(man), "count", weights.weight_count(), "oracle", eo.blob_size())
assert [(t["name"], tuple(t["shape"])) for t in man] == eo.tensor_list()
print(eo.mac_count(), sum(eo.mac_count().values()))
blob = weights.synthetic_blob()
rng = np.random.default_rng(0)
spec = (rng.integers(0, 670, size=(4, 49, 40)).astype(np.float32) * (10/256)).astype(np.float32)
o = eo.EmbeddingOracle(blob)
taps = {}
t=time.time(); e = o.forward(spec, taps); print("fwd", time.time()-t)
for k, v in taps.items():
    print(f"{k:18s} {str(v.shape):22s} mean={v.mean():+.4f} std={v.std():.4f} absmax={np.abs(v).max():.3f}")
e64 = eo.EmbeddingOracle(blob, torch.float64).forward(spec)
print("fp32 vs fp64 rel", float((e.double()-e64).abs().max() / e64.abs().max()))
