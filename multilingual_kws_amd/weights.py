"""Weight container for the embedding network + the synthetic-weight recipe used by benchmarks/tests.

The trained checkpoint the reference uses (multilingual_context_73_0.8011, docker/Dockerfile:69-70)
is a GitHub release asset and cannot be fetched here, so benchmarks and parity tests run on seeded
random weights of the same architecture.  The container is a flat float32 blob
in Keras tensor order/layout plus a JSON manifest (names = Keras variable names), so a real
checkpoint exported tensor-by-tensor drops in unchanged.
"""
import ctypes
import json
import os

import numpy as np

from . import _lib

DEFAULT_SEED = 1234


def manifest():
    """[{name, shape, offset, count}] straight from the C library (host-only call, no GPU needed)."""
    L = _lib.lib()
    n = _lib.check(L.mkws_embed_weight_manifest(None, 0))
    buf = ctypes.create_string_buffer(n + 1)
    _lib.check(L.mkws_embed_weight_manifest(buf, n + 1))
    return json.loads(buf.value.decode("utf-8"))["tensors"]


def weight_count():
    return int(_lib.lib().mkws_embed_weight_count())


def _trunc_normal(rng, shape, std):
    # Keras VarianceScaling("truncated_normal"): resample outside 2 sigma, stddev corrected by .8796
    std = std / 0.87962566103423978
    x = rng.standard_normal(shape)
    bad = np.abs(x) > 2.0
    while bad.any():
        x[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(x) > 2.0
    return (x * std).astype(np.float32)


def synthetic_blob(seed=DEFAULT_SEED, calibrate=True, tensors=None):
    """Seeded random weights following the initialisers the reference's model definition uses
    (EfficientNet conv: VarianceScaling(2, fan_out, truncated_normal); Dense: glorot_uniform,
    dense_2: lecun_normal) with non-trivial BatchNorm parameters so BN folding is exercised.
    With calibrate=True (default) the BatchNorm moving statistics are then replaced by calibrated ones (set
    from the activations of a few spectrogram-like inputs, as a trained network's would be, so every layer
    stays input-dependent and parity tests are sensitive to how data -- not just biases -- flows).  The
    calibrated statistics are DATA shipped with the package (data/synthetic_bn_<seed>.npy, written by
    tools/calibrate_synthetic_bn.py, which holds the recipe); the product package itself contains no
    host-side forward pass of the network."""
    rng = np.random.default_rng(seed)
    if tensors is None:       # (a caller without the built library -- tools/make_keras_golden.py on a TF machine -- passes tools/embedding_manifest.json)
        tensors = manifest()
    blob = np.zeros(tensors[-1]["offset"] + tensors[-1]["count"], dtype=np.float32)
    for t in tensors:
        name, shape = t["name"], tuple(t["shape"])
        leaf = name.split("/")[-1]
        if name == "normalization/mean":
            v = np.zeros(shape, np.float32)
        elif name == "normalization/variance":
            v = np.ones(shape, np.float32)
        elif leaf == "depthwise_kernel":
            v = _trunc_normal(rng, shape, np.sqrt(2.0 / (shape[0] * shape[1])))
        elif leaf == "kernel" and len(shape) == 4:
            v = _trunc_normal(rng, shape, np.sqrt(2.0 / (shape[0] * shape[1] * shape[3])))
        elif leaf == "kernel" and name.startswith("dense_2"):
            v = _trunc_normal(rng, shape, np.sqrt(1.0 / shape[0]))
        elif leaf == "kernel":
            lim = np.sqrt(6.0 / (shape[0] + shape[1]))
            v = rng.uniform(-lim, lim, shape).astype(np.float32)
        elif leaf == "bias":
            v = (0.02 * rng.standard_normal(shape)).astype(np.float32)
        elif leaf == "gamma":
            v = rng.uniform(0.5, 1.5, shape).astype(np.float32)
        elif leaf == "beta":
            v = (0.1 * rng.standard_normal(shape)).astype(np.float32)
        elif leaf == "moving_mean":
            v = (0.1 * rng.standard_normal(shape)).astype(np.float32)
        elif leaf == "moving_variance":
            v = rng.uniform(0.5, 1.5, shape).astype(np.float32)
        else:
            raise ValueError(f"no initialiser for {name}")
        blob[t["offset"]:t["offset"] + t["count"]] = v.reshape(-1)
    if calibrate:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", f"synthetic_bn_{int(seed)}.npy")
        if not os.path.exists(path):
            # only the statistics of DEFAULT_SEED ship with the package (data/): other seeds are valid weights, just with the
            # raw random BatchNorm statistics -- say so instead of failing a documented call ("synthetic:SEED")
            import warnings
            warnings.warn(f"synthetic_blob(seed={int(seed)}): no calibrated BatchNorm statistics shipped for this seed ({path}); "
                          f"using the uncalibrated ones.  `python tools/calibrate_synthetic_bn.py --seed {int(seed)}` writes them.", RuntimeWarning)
            return blob
        stats = np.load(path)
        o = 0
        for t in bn_stat_tensors(tensors):
            blob[t["offset"]:t["offset"] + t["count"]] = stats[o:o + t["count"]]
            o += t["count"]
        if o != stats.shape[0]:
            raise ValueError(f"{path} holds {stats.shape[0]} values, this architecture has {o} BatchNorm statistics")
    return blob


def bn_stat_tensors(tensors=None):
    """Manifest entries of every BatchNorm moving_mean / moving_variance, in manifest order."""
    return [t for t in (tensors or manifest()) if t["name"].split("/")[-1] in ("moving_mean", "moving_variance")]


def save(path, blob):
    """Writes <path>/weights.bin (float32 LE) + <path>/manifest.json."""
    os.makedirs(path, exist_ok=True)
    blob = np.ascontiguousarray(blob, dtype="<f4")
    if blob.shape[0] != weight_count():
        raise ValueError(f"blob has {blob.shape[0]} floats, architecture needs {weight_count()}")
    blob.tofile(os.path.join(path, "weights.bin"))
    with open(os.path.join(path, "manifest.json"), "w") as f:
        json.dump({"format": "mkws-embedding-v1", "dtype": "float32", "tensors": manifest()}, f)


def load(path):
    """Reads a container written by save(); validates it against the library's manifest."""
    with open(os.path.join(path, "manifest.json")) as f:
        man = json.load(f)
    if man.get("tensors") != manifest():
        raise ValueError(f"{path}: manifest does not match this build's embedding architecture")
    blob = np.fromfile(os.path.join(path, "weights.bin"), dtype="<f4")
    if blob.shape[0] != weight_count():
        raise ValueError(f"{path}: weights.bin has {blob.shape[0]} floats, expected {weight_count()}")
    return blob


def from_named_tensors(named):
    """{keras variable name: ndarray} -> blob (for importing a real checkpoint tensor by tensor)."""
    tensors = manifest()
    blob = np.zeros(weight_count(), dtype=np.float32)
    for t in tensors:
        if t["name"] not in named:
            raise KeyError(f"missing tensor {t['name']}")
        v = np.asarray(named[t["name"]], dtype=np.float32)
        if tuple(v.shape) != tuple(t["shape"]):
            raise ValueError(f"{t['name']}: shape {v.shape} != {tuple(t['shape'])}")
        blob[t["offset"]:t["offset"] + t["count"]] = v.reshape(-1)
    return blob
