"""Regenerates the frontend golden fixtures.  Runs ONLY in the build container (it reads
/root/reference for the tutorial-embedded WAV inputs); the fixtures it writes are committed and are
the only thing the GPU box sees.

Outputs (tests/golden/):
  tutorial_clip{0,1,2}.wav  -- INPUT data: the three 16 kHz PCM16 clips embedded (base64) in
      multilingual_kws_intro_tutorial.ipynb cell 11 (GSC "three", MSWC-en "three", MSWC-es "tres").
  frontend_real_speech.npz  -- the oracle's raw uint16 [49,40] outputs for those clips, PCAN on/off.
      (Produced by oracle/microfrontend_oracle.c, NOT by TensorFlow: TF is not installable here.)
frontend_golden.json (hand-written, not generated) holds upstream TensorFlow's own unit-test
constants and the SURVEY.md Appendix D checksums the oracle is pinned to.
"""
import base64, json, os, re, sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle.frontend_oracle import FrontendOracle  # noqa: E402

nb = json.load(open("/root/reference/multilingual_kws_intro_tutorial.ipynb"))
wavs = []
for o in nb["cells"][11]["outputs"]:
    h = o.get("data", {}).get("text/html")
    if h is None:
        continue
    h = "".join(h) if isinstance(h, list) else h
    for m in re.finditer(r"data:audio/x-wav;base64,([A-Za-z0-9+/=]+)", h):
        wavs.append(base64.b64decode(m.group(1)))
assert len(wavs) == 3
on, off = FrontendOracle(), FrontendOracle(enable_pcan=False)
res = {}
for i, w in enumerate(wavs):
    open(os.path.join(HERE, f"tutorial_clip{i}.wav"), "wb").write(w)
    n = int.from_bytes(w[40:44], "little") // 2
    pcm = np.zeros(16000, dtype=np.int16)
    pcm[: min(n, 16000)] = np.frombuffer(w[44 : 44 + 2 * n], dtype="<i2")[:16000]
    res[f"clip{i}_pcan_on"] = on.run_i16(pcm)
    res[f"clip{i}_pcan_off"] = off.run_i16(pcm)
np.savez_compressed(os.path.join(HERE, "frontend_real_speech.npz"), **res)
print("wrote", sorted(res))
