import sys, os, time
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
out = torch.empty(B, 1024, device=dev)
def timeit(n=30):
    for _ in range(5): em.forward(x, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): em.forward(x, out=out)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
modes = [int(a) for a in sys.argv[1:]] or [0, 1]
for mode in modes:
    em.set_option("gemm_lds", mode)
    got = em.forward(x)[:4].cpu().numpy()
    err = np.abs(got - ref).max() / np.abs(ref).max()
    ms = timeit()
    print(f"gemm_lds={mode}: {ms:.4f} ms/fwd  rel err {err:.2e}")
    if mode in (0, 1):
        rows = em.profile(x, reps=5)
        print("   ", " | ".join(f"{s}:{k.split('<')[0][3:11]}<{k.split('<')[1]} {ms_*1e3:.0f}" for s, k, ms_ in rows if "gemm" in k))
