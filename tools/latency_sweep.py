import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from multilingual_kws_amd import synth
from multilingual_kws_amd.embedding import input_data, transfer_learning as tl
ms = input_data.standard_microspeech_model_settings(3)
fe = input_data._frontend_for(ms, 16000)
for mb in (int(x) for x in (sys.argv[1:] or ["1", "4", "16", "64"])):
    emb, _ = tl.load_base_model("synthetic", max_batch=mb)
    a = torch.from_numpy(synth.clips_float32(mb)).cuda()
    for opts in ({}, {"fuse_block": 0}, {"fuse_block": 1}):
        for k in ("fuse_block", "fuse_front"):
            emb.set_option(k, {"fuse_block": 2, "fuse_front": 1}[k])
        for k, v in opts.items():
            emb.set_option(k, v)
        for _ in range(10):
            emb.forward(fe.forward(a))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        N = 100
        for _ in range(N):
            emb.forward(fe.forward(a)); torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / N
        print(f"max_batch={mb:3d} B={mb:3d} {str(opts):40s} {dt*1e3:.3f} ms")
