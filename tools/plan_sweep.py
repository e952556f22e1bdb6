"""Which plan should a handle of a given max_batch use?  For each batch size: graph-replayed embedding forward (what the serving
path runs) on the multi-kernel plan vs the whole-block plan, 1 lane and 4 concurrent lanes; prints ms per batch and clips/s.
    python tools/plan_sweep.py 1 8 32 64 128 256 384"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multilingual_kws_amd import weights
from multilingual_kws_amd.embedding_model import EmbeddingModel

PLANS = {"multi": dict(fuse_block=0, fuse_mid=0, fuse_back=0, fuse_pair=0, fuse_cluster=0),
         "whole": dict(fuse_block=2, fuse_mid=1, fuse_back=1, fuse_pair=0, fuse_cluster=0),
         "whole+pair": dict(fuse_block=2, fuse_mid=1, fuse_back=1, fuse_pair=1, fuse_cluster=0),
         "cluster": dict(fuse_block=2, fuse_mid=1, fuse_back=1, fuse_pair=1, fuse_cluster=1)}
if os.environ.get("PLANS"):
    PLANS = {k: v for k, v in PLANS.items() if k in os.environ["PLANS"].split(",")}
LANES = [int(x) for x in os.environ.get("LANES", "1,4").split(",")]
BIG = int(os.environ.get("BIG_TILES", "0"))       # 1: "big_tiles" option on every handle (8-clip pairs, 4-clip 4x3 workgroups)
blob = weights.synthetic_blob()
dev = torch.device("cuda:0")
for mb in (int(x) for x in (sys.argv[1:] or ["1", "8", "32", "64", "128", "256"])):
    for plan, opts in PLANS.items():
        for lanes in LANES:
            ems = [EmbeddingModel(blob, max_batch=mb) for _ in range(lanes)]
            try:
                for em in ems:
                    for k, v in opts.items():
                        if not (k == "fuse_cluster" and v == 0 and mb > 64):
                            em.set_option(k, v)
                    if BIG:
                        em.set_option("big_tiles", 1)
            except Exception as exc:
                print(f"max_batch {mb:4d} plan {plan}: not available ({exc})")
                continue
            xs = [torch.rand((mb, 49, 40), device=dev) * 26 for _ in range(lanes)]
            outs = [torch.empty((mb, 1024), device=dev) for _ in range(lanes)]
            side = [torch.cuda.Stream() for _ in range(lanes)]
            for i in range(lanes):
                ems[i].forward(xs[i], out=outs[i])
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                main = torch.cuda.current_stream()
                ems[0].forward(xs[0], out=outs[0])
                for i in range(1, lanes):
                    side[i].wait_stream(main)
                    with torch.cuda.stream(side[i]):
                        ems[i].forward(xs[i], out=outs[i])
                for i in range(1, lanes):
                    main.wait_stream(side[i])
            for _ in range(5):
                g.replay()
            torch.cuda.synchronize()
            n = 50
            t0 = time.perf_counter()
            for _ in range(n):
                g.replay()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            # latency of ONE replay with a sync after it (what a live caller sees)
            t0 = time.perf_counter()
            for _ in range(n):
                g.replay()
                torch.cuda.synchronize()
            lat = (time.perf_counter() - t0) / n
            print(f"max_batch {mb:4d} plan {plan:10s} lanes {lanes}: {dt * 1e3:7.3f} ms per replay = {lanes * mb / dt:10.0f} clips/s; synced {lat * 1e3:7.3f} ms", flush=True)
            del g, ems
