"""Regenerates tests/golden/embedding_golden.json: for the seeded synthetic weights (seed 1234) and the
first 4 synthetic clips, the ORACLE's (PyTorch-CPU fp32, oracle/efficientnet_oracle.py) embedding head
values and norms, plus checksums of the inputs so a drifting generator is caught.  Not produced by
TensorFlow/Keras (not installable here; the reference ships no checkpoint or vectors)."""
import hashlib, json, os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from multilingual_kws_amd import synth, weights          # noqa: E402
from oracle.efficientnet_oracle import EmbeddingOracle   # noqa: E402
from oracle.frontend_oracle import FrontendOracle        # noqa: E402

blob = weights.synthetic_blob(1234)
audio = synth.clips_float32(4)
spec, raw = FrontendOracle().run_batch_f32(audio, want_u16=True)
emb = EmbeddingOracle(blob).forward(spec).numpy()
out = {
    "weights_seed": 1234,
    "blob_sha1": hashlib.sha1(blob.astype("<f4").tobytes()).hexdigest(),
    "audio_sha1": hashlib.sha1(synth.clips_int16(4).astype("<i2").tobytes()).hexdigest(),
    "spec_raw_sha1": hashlib.sha1(raw.astype("<u2").tobytes()).hexdigest(),
    "embedding_first8": [[float(v) for v in row[:8]] for row in emb],
    "embedding_l2": [float(np.linalg.norm(row)) for row in emb],
    "embedding_argmax": [int(row.argmax()) for row in emb],
}
json.dump(out, open(os.path.join(HERE, "embedding_golden.json"), "w"), indent=1)
print(out["blob_sha1"], out["embedding_l2"])
