"""Recipe for the calibrated BatchNorm statistics of the synthetic benchmark weights
(multilingual_kws_amd/weights.py::synthetic_blob): writes multilingual_kws_amd/data/synthetic_bn_<seed>.npy.

Host-side weight synthesis, not a compute path and not part of the product package: the moving statistics
of every BatchNorm are set from float64 activations of 32 spectrogram-like inputs (+ a seeded perturbation) and
quantised to a 2^-12 grid, so the file does not depend on the host's floating-point library
(tests/test_host_logic.py regenerates it and compares bit for bit).

  python tools/calibrate_synthetic_bn.py [--seed 1234]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def calibrate_bn(blob, tensors, rng, n_clips=32):
    """Sets every */moving_mean and */moving_variance from float64 activations of n_clips random
    spectrogram-like inputs (+ seeded perturbation), quantised to a 2^-12 grid so the result does not
    depend on the host's floating-point library.  Host-side weight synthesis only -- not a compute path."""
    import torch
    import torch.nn.functional as F
    from multilingual_kws_amd.arch import BLOCKS
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




def calibrated_stats(seed):
    """float32 vector of every moving_mean / moving_variance (manifest order) for synthetic_blob(seed)."""
    from multilingual_kws_amd import weights
    blob = weights.synthetic_blob(seed, calibrate=False)
    tensors = weights.manifest()
    calibrate_bn(blob, tensors, np.random.default_rng(seed + 1))
    return np.concatenate([blob[t["offset"]:t["offset"] + t["count"]] for t in weights.bn_stat_tensors(tensors)])


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=1234)
    a = ap.parse_args()
    out = os.path.join(ROOT, "multilingual_kws_amd", "data", f"synthetic_bn_{a.seed}.npy")
    v = calibrated_stats(a.seed)
    np.save(out, v)
    print(f"wrote {out}: {v.shape[0]} float32 values")
