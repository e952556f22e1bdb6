"""Debug aid: gate / block taps of the 4x3-image blocks with the squeeze-excite on the 4x4x1 instruction (fuse_se4 = 1) against the 16x16x4 path and the oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multilingual_kws_amd import weights
from multilingual_kws_amd.embedding_model import EmbeddingModel
B = int(sys.argv[1]) if len(sys.argv) > 1 else 5
blob = weights.synthetic_blob()
em = EmbeddingModel(blob, max_batch=1024)
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.integers(0, 670, size=(B, 49, 40)).astype(np.float32) * np.float32(10 / 256)).cuda()
for st in ("block4b_gate", "block4b", "block5b_gate", "block6a_gate", "block6a", "dense_2"):
    em.set_option("fuse_se4", 0); a = em.tap(x, st).cpu().numpy().reshape(B, -1)
    em.set_option("fuse_se4", 1); b = em.tap(x, st).cpu().numpy().reshape(B, -1)
    d = np.abs(a - b)
    print(st, a.shape, "max |old - new|", d.max(), "rel", d.max() / np.abs(a).max(), "first bad", np.argwhere(d > 1e-4 * np.abs(a).max())[:6].tolist())
    if st.endswith("_gate"):
        print("   old", a[0, :8], "\n   new", b[0, :8], "\n   new[64:72]", b[0, 64:72], "old[64:72]", a[0, 64:72])
