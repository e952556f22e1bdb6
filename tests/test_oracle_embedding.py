"""Checks on oracle/efficientnet_oracle.py (the CPU restatement of the Keras EfficientNetB0-based
embedding model): architecture bookkeeping from SURVEY.md Appendix B, an independent float64
loop-level re-derivation of individual layers, and the committed golden embedding."""
import hashlib
import json
import os

import numpy as np
import torch

from multilingual_kws_amd import synth, weights
from oracle import efficientnet_oracle as eo
from oracle.frontend_oracle import FrontendOracle


def test_parameter_and_mac_counts():
    # Keras: 4 048 988 conv-trunk params (+ Normalization) ; 12 967 004 to dense_2 ; 32 974 496 MAC / clip
    names = dict(eo.tensor_list())
    trunk = sum(int(np.prod(s)) for n, s in names.items() if not n.startswith(("dense", "normalization")))
    assert trunk == 4048988
    assert eo.blob_size() == 12967004 + 2
    m = eo.mac_count()
    assert m == {"stem": 144000, "pointwise": 21398528, "depthwise": 1891872, "se": 627200, "dense": 8912896}
    assert sum(m.values()) == 32974496


def test_correct_pad_table():
    # Appendix B: stem ((1,1),(0,1)); 2a ((1,1),(0,1)); 3a ((2,2),(1,2)); 4a ((1,1),(1,1)); 6a ((1,2),(2,2))
    assert eo.correct_pad(49, 40, 3) == ((1, 1), (0, 1))
    assert eo.correct_pad(25, 20, 3) == ((1, 1), (0, 1))
    assert eo.correct_pad(13, 10, 5) == ((2, 2), (1, 2))
    assert eo.correct_pad(7, 5, 3) == ((1, 1), (1, 1))
    assert eo.correct_pad(4, 3, 5) == ((1, 2), (2, 2))


def _np_bn(x, w, p):   # x NHWC float64
    return (x - w[p + "/moving_mean"]) * (w[p + "/gamma"] / np.sqrt(w[p + "/moving_variance"] + 1e-3)) + w[p + "/beta"]


def _np_swish(x):
    return x / (1.0 + np.exp(-x))


def _np_mbconv(x, w, name, cin, cout, k, stride, expand):
    """One MBConv block (Keras EfficientNet, inference) with explicit numpy loops for the depthwise conv; returns
    (expand, dw, gate, out) as float64 NHWC arrays.  Independent of torch.nn.functional (the oracle's engine)."""
    p = "block" + name
    e = x
    if expand != 1:
        e = _np_swish(_np_bn(x @ w[p + "_expand_conv/kernel"][0, 0], w, p + "_expand_bn"))
    H, W = e.shape[1], e.shape[2]
    c = k // 2
    if stride == 2:          # ZeroPadding2D(correct_pad) + "valid"
        (pt, pb), (pl, pr) = (c - (1 - H % 2), c), (c - (1 - W % 2), c)
    else:                    # "same"
        (pt, pb), (pl, pr) = (c, c), (c, c)
    ep = np.pad(e, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    kd = w[p + "_dwconv/depthwise_kernel"][..., 0]           # [k,k,C]
    Ho, Wo = (H + pt + pb - k) // stride + 1, (W + pl + pr - k) // stride + 1
    d = np.zeros((x.shape[0], Ho, Wo, e.shape[-1]))
    for oh in range(Ho):
        for ow in range(Wo):
            for i in range(k):
                for j in range(k):
                    d[:, oh, ow] += ep[:, oh * stride + i, ow * stride + j] * kd[i, j]
    d = _np_swish(_np_bn(d, w, p + "_bn"))
    s = d.mean(axis=(1, 2))
    r = _np_swish(s @ w[p + "_se_reduce/kernel"][0, 0] + w[p + "_se_reduce/bias"])
    g = 1.0 / (1.0 + np.exp(-(r @ w[p + "_se_expand/kernel"][0, 0] + w[p + "_se_expand/bias"])))
    out = _np_bn((d * g[:, None, None, :]) @ w[p + "_project_conv/kernel"][0, 0], w, p + "_project_bn")
    if stride == 1 and cin == cout:
        out = out + x
    return e, d, g, out


def test_every_block_against_explicit_loops():
    """All 16 MBConv blocks (3x3 / 5x5, stride 1 / 2 with every correct_pad case, expand 1 / 6, SE, residual) chained
    through an explicit-loop float64 re-derivation: each block's expand / depthwise / gate / output taps of the oracle
    are reproduced to 1e-9, so the oracle does not rest on F.conv2d's conventions alone."""
    from multilingual_kws_amd.arch import BLOCKS
    blob = weights.synthetic_blob()
    w = {k: v.astype(np.float64) for k, v in eo.split_blob(blob).items()}
    rng = np.random.default_rng(5)
    spec = (rng.integers(0, 670, size=(2, 49, 40)).astype(np.float32) * np.float32(10 / 256))
    taps = {}
    eo.EmbeddingOracle(blob, torch.float64).forward(spec, taps)
    x = taps["stem"].astype(np.float64)
    shapes = []
    for name, cin, cout, k, stride, expand in BLOCKS:
        p = "block" + name
        e, d, g, out = _np_mbconv(x, w, name, cin, cout, k, stride, expand)
        if expand != 1:
            assert np.allclose(e, taps[p + "_expand"], rtol=1e-9, atol=1e-11), p
        assert np.allclose(d, taps[p + "_dw"], rtol=1e-9, atol=1e-11), p
        assert np.allclose(g, taps[p + "_gate"], rtol=1e-9, atol=1e-11), p
        assert np.allclose(out, taps[p], rtol=1e-9, atol=1e-11), p
        shapes.append(out.shape[1:])
        x = out                                            # chain the re-derivation, not the oracle's output
    assert shapes[0] == (25, 20, 16) and shapes[2] == (13, 10, 24) and shapes[4] == (7, 5, 40) and shapes[7] == (4, 3, 80)
    assert shapes[10] == (4, 3, 112) and shapes[14] == (2, 2, 192) and shapes[15] == (2, 2, 320)
    top = _np_swish(_np_bn(x @ w["top_conv/kernel"][0, 0], w, "top_bn"))
    assert np.allclose(top, taps["top"], rtol=1e-9, atol=1e-11)


def test_stem_and_head_against_explicit_math():
    blob = weights.synthetic_blob()
    w = {k: v.astype(np.float64) for k, v in eo.split_blob(blob).items()}
    rng = np.random.default_rng(6)
    spec = (rng.integers(0, 670, size=(1, 49, 40)).astype(np.float32) * np.float32(10 / 256))
    taps = {}
    emb = eo.EmbeddingOracle(blob, torch.float64).forward(spec, taps).numpy()
    x = np.pad(spec[0].astype(np.float64) / 255.0, ((1, 1), (0, 1)))
    k = w["stem_conv/kernel"][:, :, 0, :]
    st = np.zeros((25, 20, 32))
    for oh in range(25):
        for ow in range(20):
            st[oh, ow] = np.einsum("ij,ijc->c", x[2 * oh:2 * oh + 3, 2 * ow:2 * ow + 3], k)
    st = _np_swish(_np_bn(st, w, "stem_bn"))
    assert np.allclose(st, taps["stem"][0], rtol=1e-9, atol=1e-11)
    g = taps["top"][0].astype(np.float64).mean(axis=(0, 1))
    h = np.maximum(g @ w["dense/kernel"] + w["dense/bias"], 0)
    h = np.maximum(h @ w["dense_1/kernel"] + w["dense_1/bias"], 0)
    z = h @ w["dense_2/kernel"] + w["dense_2/bias"]
    selu = 1.0507009873554805 * np.where(z > 0, z, 1.6732632423543772 * np.expm1(z))
    assert np.allclose(selu, emb[0], rtol=1e-9, atol=1e-11)
    assert emb.shape == (1, 1024)


def test_synthetic_network_is_input_dependent_and_well_conditioned():
    blob = weights.synthetic_blob()
    spec = FrontendOracle().run_batch_f32(synth.clips_float32(3))
    e32 = eo.EmbeddingOracle(blob).forward(spec).numpy()
    e64 = eo.EmbeddingOracle(blob, torch.float64).forward(spec).numpy()
    assert np.abs(e32 - e64).max() / np.abs(e64).max() < 2e-5
    assert np.abs(e64[0] - e64[1]).mean() / np.abs(e64[0]).mean() > 0.2     # clips map to clearly different embeddings


def test_golden_embedding(golden_dir):
    G = json.load(open(os.path.join(golden_dir, "embedding_golden.json")))
    blob = weights.synthetic_blob(G["weights_seed"])
    assert hashlib.sha1(blob.astype("<f4").tobytes()).hexdigest() == G["blob_sha1"]
    assert hashlib.sha1(synth.clips_int16(4).astype("<i2").tobytes()).hexdigest() == G["audio_sha1"]
    spec, raw = FrontendOracle().run_batch_f32(synth.clips_float32(4), want_u16=True)
    assert hashlib.sha1(raw.astype("<u2").tobytes()).hexdigest() == G["spec_raw_sha1"]
    emb = eo.EmbeddingOracle(blob).forward(spec).numpy()
    assert np.allclose(emb[:, :8], np.asarray(G["embedding_first8"]), rtol=1e-4, atol=1e-6)
    assert np.allclose(np.linalg.norm(emb, axis=1), G["embedding_l2"], rtol=1e-5)
    assert [int(r.argmax()) for r in emb] == G["embedding_argmax"]
