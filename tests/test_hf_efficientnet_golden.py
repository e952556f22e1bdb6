"""The network oracle (and the HIP kernels) against a THIRD-PARTY EfficientNet-B0: Hugging Face transformers'
PyTorch port of keras/applications/efficientnet.py, run in the build container by
tests/golden/make_hf_efficientnet_golden.py on the seed-1234 weights (fixture: tests/golden/hf_efficientnet_golden.npz,
inputs and outputs only).  First check of SURVEY Appendix B that this project did not write itself; it does not replace
tests/golden/keras_golden.npz (real TensorFlow output), so DESIGN.md keeps "parity: partial".

Part A ("unit/"): the unmodified port, one block at a time, on map sizes where its size-independent padding equals
Keras' correct_pad.  Part B ("chain/"): the whole trunk on [49,40] with the five size-dependent pads re-set per axis from
the port's own correct_pad -- the values are in the fixture and are asserted against the oracle's correct_pad here."""
import os

import numpy as np
import pytest
import torch

from multilingual_kws_amd import weights
from oracle import efficientnet_oracle as eo

TOL = 2e-5          # fp32 vs fp32, different summation order (measured ~2e-6)


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(os.path.join(golden_dir, "hf_efficientnet_golden.npz"))


@pytest.fixture(scope="module")
def oracle(G):
    return eo.EmbeddingOracle(weights.synthetic_blob(int(G["weights_seed"])))


def _nchw(a):
    return torch.from_numpy(np.ascontiguousarray(a.transpose(0, 3, 1, 2)))


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().numpy()


def _rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


def test_every_block_of_the_unmodified_port(G, oracle):
    assert _rel(_nhwc(oracle.stem(_nchw(G["unit/stem/in"]))), G["unit/stem/out"]) < TOL
    seen_sizes = set()
    for block in eo.BLOCKS:
        name, cin, cout, k, s, e = block
        x = G[f"unit/block{name}/in"]
        if s == 2:
            assert x.shape[1] % 2 == 0 and x.shape[2] % 2 == 0        # the only sizes where the port pads like Keras
        got = _nhwc(oracle.mbconv(_nchw(x), block))
        ref = G[f"unit/block{name}/out"]
        assert got.shape == ref.shape, name
        assert _rel(got, ref) < TOL, (name, _rel(got, ref))
        seen_sizes.add((x.shape[1], x.shape[2], k, s))
    assert len(seen_sizes) >= 9
    top = oracle.top(_nchw(G["unit/top/in"]))
    assert _rel(_nhwc(top), G["unit/top/out"]) < TOL
    assert _rel(top.mean(dim=(2, 3)).numpy(), G["unit/top/pooled"]) < TOL


def test_chain_pads_are_keras_correct_pad(G):
    sizes = [("stem", 49, 40, 3), ("2a", 25, 20, 3), ("3a", 13, 10, 5), ("4a", 7, 5, 3), ("6a", 4, 3, 5)]
    for (name, h, w, k), (l, r, t, b) in zip(sizes, G["chain/pads"].tolist()):
        assert eo.correct_pad(h, w, k) == ((t, b), (l, r)), name


def test_whole_trunk_against_the_port(G, oracle):
    taps = {}
    oracle.forward(G["spec"], taps)
    for name in ["stem"] + ["block" + b[0] for b in eo.BLOCKS] + ["top"]:
        ref = G["chain/" + name]
        assert taps[name].shape == ref.shape, name
        assert _rel(taps[name], ref) < TOL, (name, _rel(taps[name], ref))
    assert _rel(taps["gap"], G["chain/pooled"]) < TOL


@pytest.mark.gpu
def test_device_trunk_against_the_port(G):
    """HIP kernels vs the third-party port directly (north_star tolerance 1e-3; measured ~1e-5)."""
    from multilingual_kws_amd.embedding_model import EmbeddingModel
    dev = torch.device("cuda:0")
    em = EmbeddingModel(weights.synthetic_blob(int(G["weights_seed"])), max_batch=8, device=dev)
    x = torch.from_numpy(G["spec"]).to(dev)
    worst = 0.0
    for name in ["stem"] + ["block" + b[0] for b in eo.BLOCKS] + ["top"]:
        ref = G["chain/" + name]
        got = em.tap(x, name).cpu().numpy().reshape(ref.shape)
        worst = max(worst, _rel(got, ref))
        assert _rel(got, ref) < 1e-4, (name, _rel(got, ref))
    pooled = em.tap(x, "gap").cpu().numpy().reshape(G["chain/pooled"].shape)
    assert _rel(pooled, G["chain/pooled"]) < 1e-4
    print(f"device vs HF port: worst relative error {worst:.2e}")
