"""Several one-clip handles at once, each on a stream of its own: does the cluster-chain launch (120 co-resident workgroups per window, members
spinning for their predecessors) survive neighbours?  Prints per handle: windows that matched the lone result, exchange failures, time.
    python tools/chain_concurrency_probe.py [handles] [windows] [fuse_cluster_chain]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multilingual_kws_amd import weights
from multilingual_kws_amd.embedding_model import EmbeddingModel

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
W = int(sys.argv[2]) if len(sys.argv) > 2 else 200
CHAIN = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device("cuda:0")
blob = weights.synthetic_blob()
xs = [torch.rand((1, 49, 40), device=dev) * 26 for _ in range(N)]
ems = [EmbeddingModel(blob, max_batch=1) for _ in range(N)]
for em in ems:
    em.set_option("fuse_cluster_chain", CHAIN)
refs = [em.forward(x).clone() for em, x in zip(ems, xs)]
torch.cuda.synchronize()
streams = [torch.cuda.Stream() for _ in range(N)]
outs = [[torch.empty((1, 1024), device=dev) for _ in range(W)] for _ in range(N)]
t0 = time.perf_counter()
for w in range(W):
    for k in range(N):
        with torch.cuda.stream(streams[k]):
            ems[k].forward(xs[k], out=outs[k][w])
torch.cuda.synchronize()
dt = time.perf_counter() - t0
for k in range(N):
    ok = sum(int(torch.equal(o, refs[k])) for o in outs[k])
    nan = sum(int(torch.isnan(o).any().item()) for o in outs[k])
    print(f"handle {k}: {ok} of {W} windows identical to the lone result, {nan} poisoned; chain {ems[k].get_option('fuse_cluster_chain')}, degraded {ems[k].get_option('pair_degraded')}")
print(f"{N} handles x {W} windows in {dt * 1e3:.1f} ms = {dt / W * 1e6:.1f} us per round of {N} windows")
