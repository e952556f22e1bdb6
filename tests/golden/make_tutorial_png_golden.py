"""Extracts the reference-held OUTPUT of the real TensorFlow micro-frontend that the reference ships.

Runs ONLY in the build container (it reads /root/reference); what it writes is committed data and is
the only thing the GPU box sees.

  /root/reference/multilingual_kws_intro_tutorial.ipynb, cell 13:

      settings = input_data.standard_microspeech_model_settings(label_count=1)
      for sample, ax in zip([three_gsc[0], three_mswc_en[0], three_mswc_es[0]], axes):
          spectrogram = input_data.file2spec(settings, str(sample))
          ax.imshow(spectrogram.numpy())

  i.e. `input_data.py:38-47` -> `:19-35` (TF's AudioMicrofrontend op with the op's defaults) for three
  clips, rendered by matplotlib's `imshow` (viridis, min/max normalised, nearest resampling: every one
  of the 49x40 cells is a block of identical pixels).  The three clips are the ones cell 11 plays:
  their WAV bytes are embedded in that cell's output and already sit in tests/golden/tutorial_clip{0,1,2}.wav
  (make_frontend_golden.py).  This script asserts that correspondence and writes

  tutorial_cell13.png   -- the cell's `image/png` output, byte for byte (18 KB; DATA: an output of the
                           reference, not source text)
  viridis_bytes.json    -- matplotlib's viridis colormap as imshow emits it: `(cm.viridis(i)[:3] * 255)`
                           truncated to bytes, i = 0..255 (public colormap data; two pairs of neighbouring
                           entries share a byte triple: the test compares COLOURS, not indices)
"""
import base64
import hashlib
import json
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
NB = "/root/reference/multilingual_kws_intro_tutorial.ipynb"

nb = json.load(open(NB))
c11, c13 = nb["cells"][11], nb["cells"][13]
src11, src13 = "".join(c11["source"]), "".join(c13["source"])

# cell 13 renders file2spec of exactly the three files cell 11 plays, in the same order
assert "input_data.file2spec(settings, str(sample))" in src13 and "ax.imshow(spectrogram.numpy())" in src13
assert "standard_microspeech_model_settings(label_count=1)" in src13
order13 = re.search(r"zip\(\[(.+?)\], axes\)", src13).group(1).replace(" ", "").split(",")
assert order13 == ["three_gsc[0]", "three_mswc_en[0]", "three_mswc_es[0]"], order13
order11 = re.findall(r"listen\((\w+\[0\])\)", src11)
assert order11 == order13, (order11, order13)

wavs = []
for o in c11["outputs"]:
    h = o.get("data", {}).get("text/html")
    if h is None:
        continue
    h = "".join(h) if isinstance(h, list) else h
    wavs += [base64.b64decode(m.group(1)) for m in re.finditer(r"data:audio/x-wav;base64,([A-Za-z0-9+/=]+)", h)]
assert len(wavs) == 3
for i, w in enumerate(wavs):
    have = open(os.path.join(HERE, f"tutorial_clip{i}.wav"), "rb").read()
    assert have == w, f"tutorial_clip{i}.wav is not the clip cell 11 embeds"

pngs = [o["data"]["image/png"] for o in c13["outputs"] if "image/png" in o.get("data", {})]
assert len(pngs) == 1
png = base64.b64decode("".join(pngs[0]) if isinstance(pngs[0], list) else pngs[0])
assert png[:8] == b"\x89PNG\r\n\x1a\n"
open(os.path.join(HERE, "tutorial_cell13.png"), "wb").write(png)

from matplotlib import cm  # noqa: E402

lut = [[int(v * 255) for v in cm.viridis(i)[:3]] for i in range(256)]
json.dump({"source": "matplotlib cm.viridis(i)[:3] * 255 truncated to uint8, i = 0..255", "rgb": lut},
          open(os.path.join(HERE, "viridis_bytes.json"), "w"))
print("tutorial_cell13.png", len(png), "bytes, sha1", hashlib.sha1(png).hexdigest()[:16])
