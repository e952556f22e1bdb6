"""Batch-1 live-window latency (StreamingSession: one hipGraph replay per window, 50 keyword heads) with the dense tail on gemv_kernel
(fuse_gemv = 1, shipped) and on the MFMA GEMM + split-K fold (0), alternating in one process, and the per-kernel table of both."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multilingual_kws_amd import synth, weights
from multilingual_kws_amd.embedding import batch_streaming_analysis as bsa, input_data
from multilingual_kws_amd.embedding_model import EmbeddingModel
from multilingual_kws_amd.head import Head

dev = torch.device("cuda:0")
ms = input_data.standard_microspeech_model_settings(3)
heads = [Head(max_batch=1, seed=2000 + k, device=dev) for k in range(50)]
one = torch.from_numpy(synth.clips_float32(1)).to(dev)
blob = weights.synthetic_blob()
ems = {}
for v in (1, 0):
    ems[v] = EmbeddingModel(blob, max_batch=1, device=dev)
    ems[v].set_option("fuse_gemv", v)
sess = {v: bsa.StreamingSession(embedding=ems[v], heads=heads, model_settings=ms, batch=1) for v in (1, 0)}
for rep in range(3):
    for v in (1, 0):
        for _ in range(50):
            sess[v].infer(one)
        torch.cuda.synchronize()
        n = 300
        t0 = time.perf_counter()
        for _ in range(n):
            sess[v].infer(one)
            torch.cuda.synchronize()
        print(f"fuse_gemv={v}: {(time.perf_counter() - t0) / n * 1e3:.4f} ms per window", flush=True)
x = torch.rand((1, 49, 40), device=dev) * 26
for v in (1, 0):
    prof = ems[v].profile(x, reps=20)
    print(f"fuse_gemv={v}: forward {sum(t for _, _, t in prof) * 1e3:.1f} us;", " ".join(f"{st}={t * 1e3:.1f}" for st, k, t in prof if st.startswith(("top", "dense", "gap"))))
