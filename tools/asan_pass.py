"""One pass of the embedding forward under AddressSanitizer (device-side instrumentation: mkws_embed.hip built with
-fsanitize=address for gfx950:xnack+, run with HSA_XNACK=1 and the ASan runtime preloaded -- tools/gpu/r5_asan.sh).
Prints one line per handle size; a device-side out-of-bounds access aborts the process with an ASan report."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
print("python up", flush=True)
import torch
print("torch", torch.__version__, "cuda", torch.cuda.is_available(), flush=True)
from multilingual_kws_amd import _lib, synth, weights
from multilingual_kws_amd.embedding_model import EmbeddingModel
from multilingual_kws_amd.frontend import Frontend
print("library", os.environ.get("MKWS_LIB"), "abi", _lib.lib().mkws_abi_version(), flush=True)
blob = weights.synthetic_blob()
spec64 = Frontend().forward(torch.from_numpy(synth.clips_float32(64)).cuda())
for B in [int(a) for a in sys.argv[1:]] or [1, 3, 64, 511, 512, 1024]:
    em = EmbeddingModel(blob, max_batch=B)
    spec = spec64.repeat((B + 63) // 64, 1, 1)[:B].contiguous()
    for opts in ({}, {"fuse_chain": 0}):
        for k, v in opts.items():
            em.set_option(k, v)
        out = em.forward(spec)
        torch.cuda.synchronize()
        print(f"B={B} {opts}: finite={bool(torch.isfinite(out).all())} |emb|max={float(out.abs().max()):.4f}", flush=True)
    em.close()
print("asan pass complete", flush=True)
