"""Do independent batches on separate HIP streams run CONCURRENTLY on this stack -- and does it depend on how they are submitted?
For a handle size (default 256 clips), `lanes` handles with their own workspaces, workgroup shapes of the handle's own rule or of the
1024-clip plan ("big_tiles": 4-clip workgroups / 8-clip pairs, i.e. a quarter of the chip per launch at 256 clips):
  fork    ONE hipGraph that forks into `lanes` branches (what _BatchGraph captures)
  graphs  one hipGraph per lane, each replayed on its own stream
  eager   launch by launch on `lanes` streams, round-robin from one host thread
Prints ms per round (= lanes batches) and clips/s.     python tools/lane_modes.py [max_batch] ; env LANES=1,2,4  BIG=0,1  ROUNDS=200"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multilingual_kws_amd import weights
from multilingual_kws_amd.embedding_model import EmbeddingModel

mb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
LANES = [int(x) for x in os.environ.get("LANES", "1,2,4").split(",")]
BIGS = [int(x) for x in os.environ.get("BIG", "0,1").split(",")]
ROUNDS = int(os.environ.get("ROUNDS", "200"))
MODES = os.environ.get("MODES", "fork,graphs,eager,join").split(",")
blob = weights.synthetic_blob()
dev = torch.device("cuda:0")


def timed(fn, rounds):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(rounds):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / rounds


for big in BIGS:
    for lanes in LANES:
        ems = [EmbeddingModel(blob, max_batch=mb) for _ in range(lanes)]
        if big == 1:
            for em in ems:
                em.set_option("big_tiles", 1)
        elif big == 2:                                      # the product's rule: plan for the clips all lanes hold together
            for em in ems:
                em.set_option("plan_batch", lanes * mb)
        xs = [torch.rand((mb, 49, 40), device=dev) * 26 for _ in range(lanes)]
        outs = [torch.empty((mb, 1024), device=dev) for _ in range(lanes)]
        streams = [torch.cuda.Stream(device=dev) for _ in range(lanes)]
        for i in range(lanes):
            with torch.cuda.stream(streams[i]):
                ems[i].forward(xs[i], out=outs[i])
        torch.cuda.synchronize()
        ref = [o.clone() for o in outs]
        res = {}
        if "fork" in MODES:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                main = torch.cuda.current_stream()
                ems[0].forward(xs[0], out=outs[0])
                for i in range(1, lanes):
                    streams[i].wait_stream(main)
                    with torch.cuda.stream(streams[i]):
                        ems[i].forward(xs[i], out=outs[i])
                for i in range(1, lanes):
                    main.wait_stream(streams[i])
            res["fork"] = timed(g.replay, ROUNDS)
            del g
        if "graphs" in MODES:
            gs = []
            for i in range(lanes):
                gi = torch.cuda.CUDAGraph()
                with torch.cuda.stream(streams[i]):
                    with torch.cuda.graph(gi, stream=streams[i]):
                        ems[i].forward(xs[i], out=outs[i])
                gs.append(gi)

            def all_graphs():
                for i in range(lanes):
                    with torch.cuda.stream(streams[i]):
                        gs[i].replay()
            res["graphs"] = timed(all_graphs, ROUNDS)
            del gs
        if "eager" in MODES:
            def all_eager():
                for i in range(lanes):
                    with torch.cuda.stream(streams[i]):
                        ems[i].forward(xs[i], out=outs[i])
            res["eager"] = timed(all_eager, ROUNDS)
        if "join" in MODES:
            # what a split INSIDE one forward call would do: the lanes start behind the caller's stream and the caller's stream waits for all of them
            def all_join():
                main = torch.cuda.current_stream()
                for i in range(1, lanes):
                    streams[i].wait_stream(main)
                ems[0].forward(xs[0], out=outs[0])
                for i in range(1, lanes):
                    with torch.cuda.stream(streams[i]):
                        ems[i].forward(xs[i], out=outs[i])
                for i in range(1, lanes):
                    main.wait_stream(streams[i])
            res["join"] = timed(all_join, ROUNDS)
        torch.cuda.synchronize()
        same = all(torch.equal(a, b) for a, b in zip(ref, outs))
        line = "  ".join(f"{m} {dt * 1e3:7.3f} ms = {lanes * mb / dt:9.0f} clips/s" for m, dt in res.items())
        print(f"max_batch {mb} big_tiles {big} lanes {lanes}: {line}  (outputs unchanged: {same})", flush=True)
        del ems
