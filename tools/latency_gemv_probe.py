"""Batch-1 live-window latency (StreamingSession: one hipGraph replay per window, 50 keyword heads) under the MKWS_GEMM_FORCE experiment hook
of the small-batch dense tail: the process's environment decides the tiling / split-K of every GEMM with at most Mmax planned rows.
    MKWS_GEMM_FORCE="32,1,2,8" python tools/latency_gemv_probe.py"""
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
em1 = EmbeddingModel(weights.synthetic_blob(), max_batch=1, device=dev)
heads = [Head(max_batch=1, seed=2000 + k, device=dev) for k in range(50)]
one = torch.from_numpy(synth.clips_float32(1)).to(dev)
sess = bsa.StreamingSession(embedding=em1, heads=heads, model_settings=ms, batch=1)
for rep in range(3):
    for _ in range(50):
        sess.infer(one)
    torch.cuda.synchronize()
    n = 300
    t0 = time.perf_counter()
    for _ in range(n):
        sess.infer(one)
        torch.cuda.synchronize()
    print(f"MKWS_GEMM_FORCE={os.environ.get('MKWS_GEMM_FORCE', '-')}: {(time.perf_counter() - t0) / n * 1e3:.4f} ms per window", flush=True)
