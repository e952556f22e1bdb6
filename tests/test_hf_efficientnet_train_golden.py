"""The TRAINING-MODE oracle (oracle/efficientnet_train_oracle.py: batch-statistics BatchNorm, moving-average updates, every gradient)
and the HIP training operators (`-m gpu`, EmbeddingTrainer) against a THIRD-PARTY implementation: Hugging Face transformers'
EfficientNetModel in .train(), float64, run in the build container by tests/golden/make_hf_efficientnet_train_golden.py (fixture:
inputs, loss, embedding, updated running statistics, per-tensor gradients -- whole up to 4096 entries, strided samples + L2 norm +
largest entry above).  Before this fixture the gradient oracle of SURVEY row f4 was checked only against finite differences of itself.
Reference semantics: multilingual_kws/embedding/transfer_learning.py:94-112 (Keras fit with the base model un-frozen)."""
import os

import numpy as np
import pytest
import torch

from multilingual_kws_amd import weights


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(os.path.join(golden_dir, "hf_efficientnet_train_golden.npz"))


def _sample_index(count, keep):
    return np.arange(0, count, max(1, count // keep))


def _check_gradients(G, got, tol, floor_rel):
    """got: {Keras name: array in the Keras layout}.  Every tensor against the fixture: sampled entries, L2 norm, largest entry."""
    names = [k[len("grad/"):] for k in G.files if k.startswith("grad/")]
    assert len(names) == 217
    gmax_all = max(float(G["gmax/" + n]) for n in names)
    worst = 0.0
    for n in names:
        g = np.asarray(got[n], dtype=np.float64).reshape(-1)
        ref = G["grad/" + n]
        idx = _sample_index(g.size, int(G["keep"]))
        assert idx.size == ref.size, n
        scale = max(float(G["gmax/" + n]), floor_rel * gmax_all)
        err = float(np.abs(g[idx] - ref).max()) / scale
        nerr = abs(float(np.sqrt((g * g).sum())) - float(G["gnorm/" + n])) / max(float(G["gnorm/" + n]), floor_rel * gmax_all)
        merr = abs(float(np.abs(g).max()) - float(G["gmax/" + n])) / scale
        assert err < tol and nerr < tol and merr < tol, (n, err, nerr, merr)
        worst = max(worst, err, nerr, merr)
    return worst


def test_training_oracle_matches_the_third_party_port(G):
    from oracle.efficientnet_train_oracle import TrainableEmbeddingOracle
    o = TrainableEmbeddingOracle(weights.synthetic_blob(int(G["weights_seed"])))
    o.zero_grad()
    emb = o.forward(G["spec"], training=True, drop_masks=None)
    loss = (emb * torch.from_numpy(G["R"])).sum()
    loss.backward()
    assert abs(loss.item() - float(G["loss"])) < 1e-9 * max(1.0, abs(float(G["loss"])))
    assert np.abs(emb.detach().numpy() - G["embedding"]).max() < 1e-10
    # a beta that feeds a 1x1 conv followed by a batch-statistics BN has an exactly-zero gradient: both sides return rounding noise there,
    # so tensors are compared relative to their own largest entry with a floor of 1e-6 of the network's largest gradient
    worst = _check_gradients(G, o.grads(), tol=1e-6, floor_rel=1e-6)        # float64 vs float64
    print("largest relative gradient difference oracle vs HF port:", worst)
    moving = [k[len("moving/"):] for k in G.files if k.startswith("moving/")]
    assert len(moving) == 98 and set(moving) == set(o.new_moving)
    for n in moving:        # momentum 0.99, Bessel-corrected batch variance
        assert np.abs(o.new_moving[n].numpy() - G["moving/" + n]).max() < 1e-10 * max(1.0, float(np.abs(G["moving/" + n]).max())), n


@pytest.mark.gpu
def test_hip_training_operators_match_the_third_party_port(G):
    """EmbeddingTrainer (fp32 HIP operators): gradients within 1e-3 of each tensor's largest entry, embedding 1e-4, statistics 1e-5."""
    from multilingual_kws_amd.embedding_trainer import EmbeddingTrainer
    tr = EmbeddingTrainer(weights.synthetic_blob(int(G["weights_seed"])))
    emb = tr.forward_train(torch.from_numpy(G["spec"]).cuda(), None)
    ref = G["embedding"]
    assert float(np.abs(emb.cpu().numpy() - ref).max() / np.abs(ref).max()) < 1e-4
    tr.backward(torch.from_numpy(G["R"].astype(np.float32)).cuda())
    worst = _check_gradients(G, tr.named_grads(), tol=1e-3, floor_rel=1e-3)
    print("largest relative gradient difference HIP vs HF port:", worst)
    newp = tr.blob()
    for k in G.files:
        if k.startswith("moving/"):
            t = tr.tensors[k[len("moving/"):]]
            got = newp[t["offset"]:t["offset"] + t["count"]]
            assert float(np.abs(got - G[k]).max() / max(float(np.abs(G[k]).max()), 1e-30)) < 1e-5, k
