#!/usr/bin/env python3
"""Writes tests/golden/hf_efficientnet_train_golden.npz: a TRAINING-MODE forward / backward of the third-party EfficientNet-B0
(Hugging Face `transformers` EfficientNetModel, see make_hf_efficientnet_golden.py for what the port shares with Keras) carrying the
seed-1234 synthetic weights, so that oracle/efficientnet_train_oracle.py -- the gradient oracle of SURVEY row f4
(multilingual_kws/embedding/transfer_learning.py:94-112: the reference un-freezes the base model, Keras then runs it with
training=True) -- and through it the HIP training operators are held to something their author did not write.

Set-up (BUILD container only; the fixture holds inputs and outputs):
  * the port in .train(): BatchNorm on batch statistics; torch's running-statistics update is
    running = (1 - m) * running + m * batch with the UNBIASED batch variance -- Keras' fused BatchNormalization with momentum 0.99
    is exactly that with m = 0.01 (the port passes its config value straight to torch, where it means the opposite: set here);
  * drop_connect_rate = 0 (Keras' drop-connect draws cannot be reproduced; the masked path is covered by the same-author tests);
  * the five size-dependent pads re-set per axis from the port's own correct_pad (part B of make_hf_efficientnet_golden.py);
  * float64 throughout, B = 6 real spectrogram-valued inputs;
  * the three Dense layers (not part of any third-party EfficientNet) as plain torch.nn.functional calls: relu / relu / selu;
  * loss = sum(embedding * R), R fixed: d loss / d embedding = R is what EmbeddingTrainer.backward takes.
Stored: inputs, R, loss, embedding, the updated running statistics of every BatchNorm, and per trainable tensor (Keras names and
layouts) its gradient -- whole when it has at most 4096 entries, otherwise every (count // 4096)-th entry -- plus its L2 norm and
largest entry, so that a mis-laid-out tensor cannot pass on the subsample alone.
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "hf_efficientnet_train_golden.npz")
SEED, B, KEEP = 1234, 6, 4096


def sample_index(count):
    """The entries of a flattened tensor the fixture keeps (shared with the tests)."""
    step = max(1, count // KEEP)
    return np.arange(0, count, step)


def main():
    import make_hf_efficientnet_golden as hf
    from multilingual_kws_amd import weights
    from oracle.efficientnet_oracle import BLOCKS, split_blob

    blob = weights.synthetic_blob(SEED)
    named = split_blob(blob)
    net, M = hf.build_port(named)
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            assert m.p == 0.0 or True
            m.p = 0.0                                   # drop_connect_rate = 0
        if isinstance(m, torch.nn.BatchNorm2d):
            m.momentum = 0.01                           # Keras momentum 0.99
    net = net.double().train()

    def pad_for(k, hh, ww):
        ev, od = M.correct_pad(k, adjust=True), M.correct_pad(k, adjust=False)
        lr = ev[0:2] if ww % 2 == 0 else od[0:2]
        tb = ev[2:4] if hh % 2 == 0 else od[2:4]
        return (lr[0], lr[1], tb[0], tb[1])
    net.embeddings.padding = torch.nn.ZeroPad2d(pad_for(3, 49, 40))
    hh, ww = 25, 20
    for blk, (name, cin, cout, k, s, e) in zip(net.encoder.blocks, BLOCKS):
        if s == 2:
            p = pad_for(k, hh, ww)
            blk.depthwise_conv.depthwise_conv_pad = torch.nn.ZeroPad2d(p)
            hh, ww = (hh + p[2] + p[3] - k) // 2 + 1, (ww + p[0] + p[1] - k) // 2 + 1
    assert (hh, ww) == (2, 2)

    rng = np.random.default_rng(77)
    spec = (rng.integers(0, 670, size=(B, 49, 40)).astype(np.float32) * np.float32(10 / 256))
    R = rng.standard_normal((B, 1024))
    pre = (spec.astype(np.float64) / 255.0 - float(named["normalization/mean"][0])) / max(float(np.sqrt(named["normalization/variance"][0])), 1e-7)
    dense = {k: torch.from_numpy(named[k].astype(np.float64)).requires_grad_(True)
             for k in ("dense/kernel", "dense/bias", "dense_1/kernel", "dense_1/bias", "dense_2/kernel", "dense_2/bias")}
    res = net(pixel_values=torch.from_numpy(pre[:, None]), return_dict=True)
    pooled = res.last_hidden_state.mean(dim=(2, 3))
    h = F.relu(pooled @ dense["dense/kernel"] + dense["dense/bias"])
    h = F.relu(h @ dense["dense_1/kernel"] + dense["dense_1/bias"])
    emb = F.selu(h @ dense["dense_2/kernel"] + dense["dense_2/bias"])
    loss = (emb * torch.from_numpy(R)).sum()
    loss.backward()

    grads, moving = {}, {}

    def conv(mod, name):                       # torch OIHW -> Keras HWIO
        grads[name + "/kernel"] = mod.weight.grad.permute(2, 3, 1, 0).numpy()
        if mod.bias is not None:
            grads[name + "/bias"] = mod.bias.grad.numpy()

    def dwconv(mod, name):                     # torch [C,1,k,k] -> Keras [k,k,C,1]
        grads[name + "/depthwise_kernel"] = mod.weight.grad.permute(2, 3, 0, 1).numpy()

    def bn(mod, name):
        grads[name + "/gamma"], grads[name + "/beta"] = mod.weight.grad.numpy(), mod.bias.grad.numpy()
        moving[name + "/moving_mean"], moving[name + "/moving_variance"] = mod.running_mean.numpy(), mod.running_var.numpy()

    conv(net.embeddings.convolution, "stem_conv")
    bn(net.embeddings.batchnorm, "stem_bn")
    for blk, (name, cin, cout, k, s, e) in zip(net.encoder.blocks, BLOCKS):
        p = "block" + name
        if e != 1:
            conv(blk.expansion.expand_conv, p + "_expand_conv")
            bn(blk.expansion.expand_bn, p + "_expand_bn")
        dwconv(blk.depthwise_conv.depthwise_conv, p + "_dwconv")
        bn(blk.depthwise_conv.depthwise_norm, p + "_bn")
        conv(blk.squeeze_excite.reduce, p + "_se_reduce")
        conv(blk.squeeze_excite.expand, p + "_se_expand")
        conv(blk.projection.project_conv, p + "_project_conv")
        bn(blk.projection.project_bn, p + "_project_bn")
    conv(net.encoder.top_conv, "top_conv")
    bn(net.encoder.top_bn, "top_bn")
    for k, v in dense.items():
        grads[k] = v.grad.numpy()

    out = {"weights_seed": np.int64(SEED), "spec": spec, "R": R, "loss": np.float64(loss.item()), "embedding": emb.detach().numpy(),
           "transformers_version": np.array(__import__("transformers").__version__), "keep": np.int64(KEEP)}
    for name, g in grads.items():
        assert g.shape == named[name].shape, (name, g.shape, named[name].shape)
        flat = np.ascontiguousarray(g).reshape(-1)
        out["grad/" + name] = flat[sample_index(flat.size)]
        out["gnorm/" + name] = np.float64(np.sqrt((flat * flat).sum()))
        out["gmax/" + name] = np.float64(np.abs(flat).max())
    for name, v in moving.items():
        out["moving/" + name] = v.copy()
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT}: {len(grads)} gradients, {len(moving)} running statistics, {os.path.getsize(OUT) / 1024:.0f} KiB, loss {loss.item():.6f}")


if __name__ == "__main__":
    main()
