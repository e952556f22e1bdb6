"""Workload for `rocprofv3 --kernel-trace --stats`: the batch-1 serving chain (frontend -> embedding -> 50 heads), 300 graph replays
with a synchronisation after each -- which kernels make up the 0.5 ms a live window costs?"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multilingual_kws_amd import synth, weights
from multilingual_kws_amd.embedding import batch_streaming_analysis as bsa, input_data
from multilingual_kws_amd.embedding_model import EmbeddingModel
from multilingual_kws_amd.head import Head

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ms = input_data.standard_microspeech_model_settings(3)
em = EmbeddingModel(weights.synthetic_blob(), max_batch=B)
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    em.set_option(k, int(v))
heads = [Head(max_batch=B, seed=2000 + k) for k in range(50)]
sess = bsa.StreamingSession(embedding=em, heads=heads, model_settings=ms, batch=B)
a = torch.from_numpy(synth.clips_float32(B)).cuda()
for _ in range(300):
    sess.infer(a)
    torch.cuda.synchronize()
