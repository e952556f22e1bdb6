"""Does the micro-frontend of batch k+1 (integer VALU, no MFMA) hide under the embedding of batch k when they run on two streams?
python tools/overlap_probe.py -- ms per step: serial (one stream) vs pipelined (frontend one batch ahead on a second stream)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multilingual_kws_amd import synth, weights
from multilingual_kws_amd.embedding_model import EmbeddingModel
from multilingual_kws_amd.frontend import Frontend
B = 1024
dev = torch.device("cuda:0")
em = EmbeddingModel(weights.synthetic_blob(), max_batch=B)
fe = Frontend(max_samples=16000)
audio = torch.from_numpy(synth.clips_float32(B)).to(dev)
specs = [torch.empty((B, 49, 40), device=dev) for _ in range(2)]
emb = torch.empty((B, 1024), device=dev)
s_fe, s_em = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
def serial(n):
    for i in range(n):
        fe.forward(audio, out=specs[0]); em.forward(specs[0], out=emb)
def pipelined(n):
    ev_fe = [torch.cuda.Event() for _ in range(2)]; ev_em = [torch.cuda.Event() for _ in range(2)]
    with torch.cuda.stream(s_fe):
        fe.forward(audio, out=specs[0]); ev_fe[0].record(s_fe)
    for i in range(n):
        cur, nxt = i & 1, (i + 1) & 1
        with torch.cuda.stream(s_fe):
            if i >= 1: s_fe.wait_event(ev_em[nxt])          # the embedding that read specs[nxt] (step i-1) must be done
            fe.forward(audio, out=specs[nxt]); ev_fe[nxt].record(s_fe)
        with torch.cuda.stream(s_em):
            s_em.wait_event(ev_fe[cur])
            em.forward(specs[cur], out=emb); ev_em[cur].record(s_em)
def timeit(fn, n=300):
    fn(20); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(n); torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for rep in range(3):
    print(f"serial {timeit(serial):.4f} ms/step   pipelined {timeit(pipelined):.4f} ms/step")
ref = emb.clone(); serial(1); torch.cuda.synchronize(); print("same result:", torch.equal(ref, emb))
