import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from multilingual_kws_amd import weights
from multilingual_kws_amd.embedding_model import EmbeddingModel
from multilingual_kws_amd.head import Head
from multilingual_kws_amd.frontend import Frontend
from oracle import efficientnet_oracle as eo
from oracle import head_oracle as ho
dev = torch.device("cuda:0")
blob = weights.synthetic_blob()
rng = np.random.default_rng(0)
spec = (rng.integers(0, 670, size=(6, 49, 40)).astype(np.float32) * np.float32(10/256))
o = eo.EmbeddingOracle(blob)
taps = {}
e_ref = o.forward(spec, taps).numpy()
em = EmbeddingModel(blob, max_batch=1024)
x = torch.from_numpy(spec).to(dev)
worst = 0
for k, v in taps.items():
    try:
        got = em.tap(x, k).cpu().numpy().reshape(v.shape)
    except Exception as ex:
        print(k, "ERR", ex); continue
    err = np.abs(got - v).max() / (np.abs(v).max() + 1e-12)
    worst = max(worst, err)
    flag = "" if err < 1e-4 else "  <<<<<"
    print(f"{k:18s} rel_err={err:.3e}{flag}")
e = em.forward(x).cpu().numpy()
print("embedding rel err", np.abs(e - e_ref).max() / np.abs(e_ref).max(), "worst tap", worst)
# timing
B = 1024
xs = torch.from_numpy((rng.integers(0, 670, size=(B, 49, 40)).astype(np.float32) * np.float32(10/256))).to(dev)
out = torch.empty(B, 1024, device=dev)
for _ in range(3): em.forward(xs, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): em.forward(xs, out=out)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"embed B=1024: {ms:.3f} ms/batch, {B/ms*1000:.0f} clips/s, {B*65.95e6/ms/1e9:.1f} TFLOP/s")
# head
p0 = ho.glorot_uniform_params(seed=3)
hd = Head(params=p0, max_batch=1024)
embs = torch.from_numpy(rng.standard_normal((300, 1024)).astype(np.float32) * 0.3).to(dev)
ys = torch.from_numpy(rng.integers(0, 3, size=300).astype(np.int32)).to(dev)
probs = hd.forward(embs).cpu().numpy()
pr, _ = ho.forward(p0, embs.cpu().numpy())
print("head probs err", np.abs(probs - pr).max())
st = hd.loss_grad(embs, ys).tolist()
loss, g, ncorr, lsum = ho.loss_and_grad(p0, embs.cpu().numpy(), ys.cpu().numpy())
gg = hd.grad_view().cpu().numpy()
print("head loss", st[0], lsum, "correct", st[1], ncorr, "grad rel err", np.abs(gg - g).max() / np.abs(g).max())
opt = ho.KerasAdam(len(p0)); p = p0.astype(np.float64)
for t in range(5):
    hd.loss_grad(embs, ys); hd.adam_step(lr=1e-3)
    _, g, _, _ = ho.loss_and_grad(p, embs.cpu().numpy(), ys.cpu().numpy()); p = opt.step(p, g)
print("head adam 5 steps param err", np.abs(hd.get_params() - p).max(), "max |dp|", np.abs(p - p0).max())
