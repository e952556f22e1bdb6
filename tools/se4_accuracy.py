"""Element-wise error of the 1024-clip embedding against the fp32 PyTorch-CPU oracle and an fp64 oracle run, with the squeeze-excite of the 4x3-image
blocks on the 4x4x1 instruction (fuse_se4 = 1) and on the 16x16x4 streams (0): worst |a - b| - 1e-3 |b| in units of max |b| (the test's floor is 1e-5)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multilingual_kws_amd import weights
from multilingual_kws_amd.embedding_model import EmbeddingModel
from oracle.efficientnet_oracle import EmbeddingOracle
blob = weights.synthetic_blob()
em = EmbeddingModel(blob, max_batch=1024)
orc = EmbeddingOracle(blob)
rng = np.random.default_rng(2)
spec = (rng.integers(0, 670, size=(1024, 49, 40)).astype(np.float32) * np.float32(10 / 256))
x = torch.from_numpy(spec).cuda()
ref = np.concatenate([orc.forward(spec[s:s + 128]).numpy() for s in range(0, 1024, 128)])
try:
    orc64 = EmbeddingOracle(blob, dtype=torch.float64)
    ref64 = np.concatenate([orc64.forward(spec[s:s + 128]).numpy() for s in range(0, 256, 128)]).astype(np.float64)
except Exception as e:
    ref64 = None
    print("no fp64 oracle:", e)
for v in (0, 1, 0, 1):
    em.set_option("fuse_se4", v)
    got = em.forward(x).cpu().numpy()
    m = np.abs(ref).max()
    ex = (np.abs(got - ref) - 1e-3 * np.abs(ref)) / m
    line = f"fuse_se4={v}: vs fp32 oracle: max |a-b|/max|b| {np.abs(got - ref).max() / m:.3e}, worst excess over 1e-3|b| {ex.max():.3e} of max|b|, elements over the 1e-5 floor {(ex > 1e-5).sum()}"
    if ref64 is not None:
        m64 = np.abs(ref64).max()
        line += f"; vs fp64 oracle (256 clips): GPU {np.abs(got[:256] - ref64).max() / m64:.3e}, fp32 oracle itself {np.abs(ref[:256] - ref64).max() / m64:.3e}"
    print(line)
