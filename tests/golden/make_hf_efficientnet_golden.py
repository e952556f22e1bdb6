#!/usr/bin/env python3
"""Writes tests/golden/hf_efficientnet_golden.npz: outputs of a THIRD-PARTY EfficientNet-B0 -- Hugging Face
`transformers.models.efficientnet.EfficientNetModel`, a PyTorch port of keras/applications/efficientnet.py written by
people who are not this project -- carrying the seed-1234 synthetic weights, so that oracle/efficientnet_oracle.py (and,
through it, the HIP kernels) is checked against something its author did not write.

Runs in the BUILD container only (transformers 5.x from the offline wheelhouse on the torch already present); nothing
of `transformers` travels: the fixture holds inputs and outputs, the tests read the .npz.  Like
tests/golden/make_detector_golden.py this script is the committed recipe of a committed fixture.

What the port and Keras share, and where they differ (read from modeling_efficientnet.py, transformers 5.15.0):
  * same per-block graph: expand 1x1 + BN + swish, depthwise + BN + swish, SE (width max(1, int(cin * 0.25)), swish /
    sigmoid, conv biases), project 1x1 + BN, residual on stride-1 repeats; BN eps from the config (1e-3 here);
  * stride-1 depthwise: torch padding="same" == Keras "same" (odd kernels, symmetric);
  * stride-2 depthwise: ZeroPad2d(correct_pad(k, adjust=True)) = (k//2 - 1, k//2) on BOTH axes, whatever the map size.
    Keras' correct_pad gives that for even sizes only and (k//2, k//2) for odd sizes;
  * stem: ZeroPad2d((0, 1, 0, 1)), again Keras' values for even sizes only;
  * no Rescaling / Normalization layers inside the model (its image processor does that): applied here in numpy.
So the fixture has two parts:
  A. "unit/...": every block of the UNMODIFIED port run alone on an input whose spatial size makes its padding equal
     Keras' (all 11 stride-1 blocks on the real map sizes; the stem and the four stride-2 blocks on even-sized crops).
     Inputs are realistic activations (the oracle's own taps), but any input would do: the test feeds the same array to
     the oracle's block function.
  B. "chain/...": the whole trunk on real [49,40] spectrograms with the five size-dependent pads re-set, per axis, from
     the port's OWN correct_pad(k, adjust = axis size is even) -- adjust=False is its symmetric form, which is Keras'
     rule for odd sizes.  Stem + all 16 block outputs + top activation + pooled features.  This part is what the
     `-m gpu` test compares device taps with.
The three Dense layers are not part of any third-party EfficientNet; they stay pinned by the explicit float64 math of
tests/test_oracle_embedding.py::test_stem_and_head_against_explicit_math.

Keras semantics the result stands for: multilingual_kws/train_multilingual_embedding.py:58-83 (reference repo).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden", "hf_efficientnet_golden.npz")
SEED = 1234


def build_port(named):
    """EfficientNetModel (B0 geometry, 1 input channel, BN eps 1e-3) with the Keras-named tensors assigned."""
    from transformers import EfficientNetConfig, EfficientNetModel
    from transformers.models.efficientnet import modeling_efficientnet as M
    cfg = EfficientNetConfig(num_channels=1, image_size=49, width_coefficient=1.0, depth_coefficient=1.0,
                             hidden_dim=1280, batch_norm_eps=1e-3, depthwise_padding=[])
    net = EfficientNetModel(cfg).eval()

    def t(a):
        return torch.from_numpy(np.ascontiguousarray(a))

    def conv(mod, name):                       # Keras HWIO -> torch OIHW
        with torch.no_grad():
            mod.weight.copy_(t(named[name + "/kernel"].transpose(3, 2, 0, 1)))
            if mod.bias is not None:
                mod.bias.copy_(t(named[name + "/bias"]))

    def dwconv(mod, name):                     # Keras [k,k,C,1] -> torch [C,1,k,k]
        with torch.no_grad():
            mod.weight.copy_(t(named[name + "/depthwise_kernel"].transpose(2, 3, 0, 1)))

    def bn(mod, name):
        with torch.no_grad():
            mod.weight.copy_(t(named[name + "/gamma"]))
            mod.bias.copy_(t(named[name + "/beta"]))
            mod.running_mean.copy_(t(named[name + "/moving_mean"]))
            mod.running_var.copy_(t(named[name + "/moving_variance"]))

    from oracle.efficientnet_oracle import BLOCKS
    conv(net.embeddings.convolution, "stem_conv")
    bn(net.embeddings.batchnorm, "stem_bn")
    assert len(net.encoder.blocks) == len(BLOCKS)
    for blk, (name, cin, cout, k, s, e) in zip(net.encoder.blocks, BLOCKS):
        p = "block" + name
        if e != 1:
            conv(blk.expansion.expand_conv, p + "_expand_conv")
            bn(blk.expansion.expand_bn, p + "_expand_bn")
        else:
            assert not blk.expand
        dwconv(blk.depthwise_conv.depthwise_conv, p + "_dwconv")
        bn(blk.depthwise_conv.depthwise_norm, p + "_bn")
        conv(blk.squeeze_excite.reduce, p + "_se_reduce")
        conv(blk.squeeze_excite.expand, p + "_se_expand")
        conv(blk.projection.project_conv, p + "_project_conv")
        bn(blk.projection.project_bn, p + "_project_bn")
        assert blk.depthwise_conv.stride == s and blk.depthwise_conv.depthwise_conv.kernel_size == (k, k)
    conv(net.encoder.top_conv, "top_conv")
    bn(net.encoder.top_bn, "top_bn")
    return net, M


def nchw(a):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a, np.float32).transpose(0, 3, 1, 2)))


def nhwc(t):
    return t.detach().permute(0, 2, 3, 1).contiguous().numpy()


def main():
    from multilingual_kws_amd import synth, weights
    from oracle.efficientnet_oracle import BLOCKS, EmbeddingOracle, split_blob
    from oracle.frontend_oracle import FrontendOracle

    blob = weights.synthetic_blob(SEED)
    named = split_blob(blob)
    net, M = build_port(named)
    spec = FrontendOracle().run_batch_f32(synth.clips_float32(2)).astype(np.float32)       # [2,49,40], values k*10/256
    pre = (spec / np.float32(255.0) - named["normalization/mean"][0]) / max(float(np.sqrt(named["normalization/variance"][0])), 1e-7)
    pre = pre.astype(np.float32)[..., None]                                                 # NHWC, C = 1
    taps = {}
    EmbeddingOracle(blob).forward(spec, taps)            # only a source of realistic block inputs for part A
    out = {"weights_seed": np.int64(SEED), "spec": spec, "transformers_version": np.array(__import__("transformers").__version__)}

    with torch.no_grad():
        # ---- A: unmodified port, block by block, on sizes where its fixed padding is Keras' ------------------
        x = pre[:, :48, :, :]                                                              # stem on an even-sized crop
        out["unit/stem/in"] = x
        out["unit/stem/out"] = nhwc(net.embeddings(nchw(x)))
        prev = "stem"
        for blk, (name, cin, cout, k, s, e) in zip(net.encoder.blocks, BLOCKS):
            x = taps[prev]
            if s == 2:                                                                     # crop to even H and W
                x = x[:, :x.shape[1] // 2 * 2, :x.shape[2] // 2 * 2, :]
            out[f"unit/block{name}/in"] = np.ascontiguousarray(x)
            out[f"unit/block{name}/out"] = nhwc(blk(nchw(x)))
            prev = "block" + name
        x = taps["block7a"]
        out["unit/top/in"] = x
        h = net.encoder.top_activation(net.encoder.top_bn(net.encoder.top_conv(nchw(x))))
        out["unit/top/out"] = nhwc(h)
        out["unit/top/pooled"] = net.pooler(h).reshape(h.shape[:2]).numpy()

        # ---- B: whole trunk on [49,40], size-dependent pads re-set from the port's own correct_pad per axis ---
        def pad_for(k, hh, ww):
            ev = M.correct_pad(k, adjust=True)           # (left, right, top, bottom) for even sizes
            od = M.correct_pad(k, adjust=False)          # symmetric form = Keras' rule for odd sizes
            lr = ev[0:2] if ww % 2 == 0 else od[0:2]
            tb = ev[2:4] if hh % 2 == 0 else od[2:4]
            return (lr[0], lr[1], tb[0], tb[1])
        hh, ww = 49, 40
        net.embeddings.padding = torch.nn.ZeroPad2d(pad_for(3, hh, ww))
        pads = {"stem": pad_for(3, hh, ww)}
        hh, ww = 25, 20
        for blk, (name, cin, cout, k, s, e) in zip(net.encoder.blocks, BLOCKS):
            if s == 2:
                pads["block" + name] = pad_for(k, hh, ww)
                blk.depthwise_conv.depthwise_conv_pad = torch.nn.ZeroPad2d(pads["block" + name])
                l, r, tp, bt = pads["block" + name]
                hh, ww = (hh + tp + bt - k) // 2 + 1, (ww + l + r - k) // 2 + 1
        assert (hh, ww) == (2, 2)
        res = net(pixel_values=nchw(pre), output_hidden_states=True, return_dict=True)
        hs = res.hidden_states
        assert len(hs) == 17
        out["chain/stem"] = nhwc(hs[0])
        for (name, *_), h in zip(BLOCKS, hs[1:]):
            out["chain/block" + name] = nhwc(h)
        out["chain/top"] = nhwc(res.last_hidden_state)
        out["chain/pooled"] = res.pooler_output.numpy()
        out["chain/pads"] = np.array([[*pads[k]] for k in ("stem", "block2a", "block3a", "block4a", "block6a")], np.int64)

    out = {k: (v.astype(np.float32) if isinstance(v, np.ndarray) and v.dtype == np.float64 else v) for k, v in out.items()}
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT}: {len(out)} arrays, {os.path.getsize(OUT) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
