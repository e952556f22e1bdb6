import sys, os, time, tempfile
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from tests.util_data import make_fewshot_dataset
from multilingual_kws_amd.embedding import input_data, transfer_learning as tl
from multilingual_kws_amd import parallel
from multilingual_kws_amd.head import Head, glorot_uniform_params
d = make_fewshot_dataset(tempfile.mkdtemp(), n_unknown=256)
ms = input_data.standard_microspeech_model_settings(3)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
emb, blob = tl.load_base_model("synthetic", max_batch=B)
head = Head(max_batch=B, seed=0)
ds = input_data.AudioDataset(ms, ["target"], d["bg_dir"], d["unknown"], unknown_percentage=50.0, spec_aug_params=input_data.SpecAugParams(percentage=80), seed=1)
train = ds.init_single_target(input_data.AUTOTUNE, d["train"], is_training=True).shuffle(1000).repeat().batch(B)
it = iter(train)
def sync(): torch.cuda.synchronize()
for _ in range(3):
    spec, lab = next(it); e = emb.forward(spec); parallel.dp_step(head, e, lab, lr=1e-3)
sync()
N = 30
t0 = time.perf_counter(); tb = te = th = 0.0
for _ in range(N):
    a = time.perf_counter(); spec, lab = next(it); sync(); b = time.perf_counter()
    e = emb.forward(spec); sync(); c = time.perf_counter()
    parallel.dp_step(head, e, lab, lr=1e-3); sync(); dd = time.perf_counter()
    tb += b - a; te += c - b; th += dd - c
tot = time.perf_counter() - t0
print(f"B={B}: {tot/N*1e3:.2f} ms/step -> {B*N/tot:.0f} clips/s  [batch assembly+frontend {tb/N*1e3:.2f} ms, embedding {te/N*1e3:.2f} ms, head step {th/N*1e3:.2f} ms]")
# without per-phase syncs
sync(); t0 = time.perf_counter()
for _ in range(N):
    spec, lab = next(it); e = emb.forward(spec); parallel.dp_step(head, e, lab, lr=1e-3)
sync(); tot = time.perf_counter() - t0
print(f"B={B} async: {tot/N*1e3:.2f} ms/step -> {B*N/tot:.0f} clips/s")
