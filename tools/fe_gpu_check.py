import sys, time, ctypes, hashlib, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from multilingual_kws_amd import _lib
# lenient binding for this early check
L = ctypes.CDLL(_lib.LIB_PATH)
for name, res, args in _lib.SYMBOLS:
    if hasattr(L, name):
        fn = getattr(L, name); fn.restype = res; fn.argtypes = args
_lib._lib = L
from multilingual_kws_amd.frontend import Frontend
from oracle.frontend_oracle import FrontendOracle
from tests.util_signals import d3_inputs, read_wav_pcm16
dev = torch.device("cuda:0")
print(torch.cuda.get_device_name(0))
ok = True
for pcan in (1, 0):
    fe = Frontend(enable_pcan=pcan); fo = FrontendOracle(enable_pcan=bool(pcan))
    sigs = list(d3_inputs().items())
    for i in range(3):
        sigs.append((f"clip{i}", read_wav_pcm16(f"tests/golden/tutorial_clip{i}.wav")[0]))
    rng = np.random.default_rng(1)
    for i in range(8):
        amp = [30, 300, 3000, 30000, 32767, 1, 10000, 20000][i]
        sigs.append((f"rand{i}", rng.integers(-amp, amp + 1, size=16000).astype(np.int16)))
    sigs.append(("minfull", np.full(16000, -32768, dtype=np.int16)))
    sigs.append(("maxfull", np.full(16000, 32767, dtype=np.int16)))
    pcm = np.stack([s for _, s in sigs])
    exp = np.stack([fo.run_i16(s) for s in pcm])
    a16 = torch.from_numpy(pcm).to(dev)
    spec, raw = fe.forward(a16, want_raw=True)
    got = raw.cpu().numpy().view(np.uint16)
    for j, (name, _) in enumerate(sigs):
        same = np.array_equal(got[j], exp[j])
        if not same:
            ok = False
            d = np.argwhere(got[j] != exp[j])
            print("MISMATCH", pcan, name, len(d), d[:3].tolist(), got[j][tuple(d[0])], exp[j][tuple(d[0])])
    af = torch.from_numpy(pcm.astype(np.float32) / 32768.0).to(dev)
    spec2, raw2 = fe.forward(af, want_raw=True)
    ok &= bool(torch.equal(raw2, raw)) and bool(torch.equal(spec2, spec))
    ok &= np.array_equal(spec.cpu().numpy(), exp.astype(np.float32) * np.float32(10 / 256))
    print("pcan", pcan, "parity", ok)
# timing at B=1024
fe = Frontend()
B = 1024
a = (torch.rand(B, 16000, device=dev) * 1.6 - 0.8)
out = torch.empty(B, 49, 40, device=dev)
for _ in range(5): fe.forward(a, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): fe.forward(a, out=out)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 50
print(f"frontend B=1024: {ms*1000:.1f} us/batch, {B/ms*1000:.0f} clips/s, {B*71840/ms/1e6:.1f} GB/s algorithmic")
exp = FrontendOracle().run_batch_f32(a[:32].cpu().numpy())
print("rand f32 parity", np.array_equal(out[:32].cpu().numpy(), exp))
# streaming
n = 16000 * 5
s = (torch.rand(n, device=dev) * 1.6 - 0.8)
fe5 = Frontend(max_samples=n)
sp = fe5.stream(s, 16000, 320)
wins = torch.stack([s[w*320: w*320+16000] for w in range(sp.shape[0])])
ref = fe5.forward(wins)
print("stream", tuple(sp.shape), bool(torch.equal(sp, ref)))
print("ALL_OK" if ok else "FAILED")
