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


def synthetic_blob(seed=DEFAULT_SEED, calibrate=True):
    """Seeded random weights following the initialisers the reference's model definition uses
    (EfficientNet conv: VarianceScaling(2, fan_out, truncated_normal); Dense: glorot_uniform,
    dense_2: lecun_normal) with non-trivial BatchNorm parameters so BN folding is exercised.
    With calibrate=True (default) the BatchNorm moving statistics are then set from the activations of
    a few spectrogram-like calibration inputs (as a trained network's would be), so every layer stays
    input-dependent and parity tests are sensitive to how data -- not just biases -- flows."""
    rng = np.random.default_rng(seed)
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
        _calibrate_bn(blob, tensors, np.random.default_rng(seed + 1))
    return blob


def _calibrate_bn(blob, tensors, rng, n_clips=32):
    """Sets every */moving_mean and */moving_variance from float64 activations of n_clips random
    spectrogram-like inputs (+ seeded perturbation), quantised to a 2^-12 grid so the result does not
    depend on the host's floating-point library.  Host-side weight synthesis only -- not a compute path."""
    import torch
    import torch.nn.functional as F
    from .arch import BLOCKS
    T = {t["name"]: t for t in tensors}

    def get(name):
        t = T[name]
        return torch.from_numpy(blob[t["offset"]:t["offset"] + t["count"]].reshape(t["shape"]).astype(np.float64))

    def put(name, v):
        t = T[name]
        blob[t["offset"]:t["offset"] + t["count"]] = np.asarray(v, dtype=np.float32).reshape(-1)

    def bn(x, p):
        c = x.shape[1]
        mean = x.mean(dim=(0, 2, 3)).numpy()
        var = x.var(dim=(0, 2, 3), unbiased=False).numpy()
        var = np.maximum(var, 0.1 * var.mean()) + 1e-5     # no near-dead channels: bounded gain
        mean = mean + 0.1 * np.sqrt(var) * rng.standard_normal(c)
        var = var * rng.uniform(0.7, 1.4, c)
        mean = np.round(mean * 4096.0) / 4096.0
        var = np.maximum(np.round(var * 4096.0), 1.0) / 4096.0
        put(p + "/moving_mean", mean)
        put(p + "/moving_variance", var)
        g, b = get(p + "/gamma"), get(p + "/beta")
        m, v = torch.from_numpy(mean), torch.from_numpy(var)
        return (x - m.view(1, -1, 1, 1)) * (g / torch.sqrt(v + 1e-3)).view(1, -1, 1, 1) + b.view(1, -1, 1, 1)

    def conv(x, name, stride=1, bias=None):
        return F.conv2d(x, get(name).permute(3, 2, 0, 1).contiguous(), None if bias is None else get(bias), stride=stride)

    def swish(x):
        return x * torch.sigmoid(x)

    # spectrogram-like calibration fields k * 10/256, k in [0, 670]: white, smooth, banded (tonal), sparse
    q = n_clips // 4
    white = rng.uniform(0, 1, (q, 1, 49, 40))
    smooth = F.interpolate(torch.from_numpy(rng.uniform(0, 1, (q, 1, 13, 10))), size=(49, 40), mode="bilinear",
                           align_corners=False).numpy()
    bands = np.clip(0.15 * rng.uniform(0, 1, (q, 1, 49, 40)) + (rng.uniform(0, 1, (q, 1, 1, 40)) > 0.8) * rng.uniform(0.5, 1, (q, 1, 49, 1)), 0, 1)
    sparse = rng.uniform(0, 1, (n_clips - 3 * q, 1, 49, 40)) * (rng.uniform(0, 1, (n_clips - 3 * q, 1, 49, 40)) > 0.7)
    x = torch.from_numpy(np.round(np.concatenate([white, smooth, bands, sparse]) * 670.0) * (10.0 / 256.0))
    with torch.no_grad():
        x = x / 255.0
        x = F.pad(x, (0, 1, 1, 1))
        x = swish(bn(conv(x, "stem_conv/kernel", 2), "stem_bn"))
        for name, cin, cout, k, s, e in BLOCKS:
            p = "block" + name
            inp = x
            if e != 1:
                x = swish(bn(conv(x, p + "_expand_conv/kernel"), p + "_expand_bn"))
            c = k // 2
            pad = (c - (1 - x.shape[3] % 2), c, c - (1 - x.shape[2] % 2), c) if s == 2 else (c, c, c, c)
            dw = get(p + "_dwconv/depthwise_kernel").permute(2, 3, 0, 1).contiguous()
            x = swish(bn(F.conv2d(F.pad(x, pad), dw, stride=s, groups=x.shape[1]), p + "_bn"))
            se = x.mean(dim=(2, 3), keepdim=True)
            se = swish(conv(se, p + "_se_reduce/kernel", bias=p + "_se_reduce/bias"))
            x = x * torch.sigmoid(conv(se, p + "_se_expand/kernel", bias=p + "_se_expand/bias"))
            x = bn(conv(x, p + "_project_conv/kernel"), p + "_project_bn")
            if s == 1 and cin == cout:
                x = x + inp
        bn(conv(x, "top_conv/kernel"), "top_bn")


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
