"""Parity with the REAL TensorFlow / Keras code path of the reference, through tests/golden/keras_golden.npz
(written by tools/make_keras_golden.py on a machine that has TensorFlow).

The MI355X image cannot install TensorFlow, so until somebody runs that script and commits its output these tests SKIP
-- loudly: the embedding / head oracles are then pinned only by the independent float64 re-derivations and finite
differences of tests/test_oracle_*.py ("parity unpinned by the reference").  Once the file
exists, the CPU tests pin the oracles to TF bit-for-bit (frontend) / to fp32 round-off (network, head), and the
-m gpu tests pin the HIP kernels to TF directly."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "tests", "golden", "keras_golden.npz")
ORACLE_TAP = {"stem_activation": "stem", "block1a_project_bn": "block1a", "block2b_add": "block2b", "block3b_add": "block3b", "block4c_add": "block4c",
              "block5c_add": "block5c", "block6d_add": "block6d", "block7a_project_bn": "block7a", "top_activation": "top", "dense_2": "dense_2"}


@pytest.fixture(scope="module")
def K():
    if not os.path.exists(PATH):
        pytest.skip("tests/golden/keras_golden.npz is ABSENT: parity with TensorFlow/Keras is UNPINNED. "
                    "Run `python tools/make_keras_golden.py` where TensorFlow is installed and commit the file.")
    return np.load(PATH)


def test_tool_is_self_consistent_without_tensorflow():
    """The generator's TF-free pieces (signals, WAV reader, manifest, synthetic blob) run here, so the script is not dead code."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("make_keras_golden", os.path.join(ROOT, "tools", "make_keras_golden.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    from multilingual_kws_amd import weights
    from tests.util_signals import d3_inputs
    sig = tool.d3_signals()
    ref = d3_inputs()
    for k in ("square4", "sine1k", "lcg", "zeros"):
        assert np.array_equal(sig[k], ref[k]), k
    assert tool.read_wav_pcm16(os.path.join(ROOT, "tests", "golden", "tutorial_clip1.wav")).shape == (16000,)
    tensors = json.load(open(os.path.join(ROOT, "tools", "embedding_manifest.json")))["tensors"]
    assert tensors == weights.manifest()                                         # the committed dump is current
    assert np.array_equal(weights.synthetic_blob(1234, tensors=tensors), weights.synthetic_blob())


def test_frontend_oracle_equals_tensorflow(K):
    from oracle.frontend_oracle import FrontendOracle
    fo = FrontendOracle()
    names = [k.split("/", 1)[1] for k in K.files if k.startswith("fe_raw/")]
    assert len(names) >= 7
    for n in names:
        pcm = K[f"fe_in/{n}"]
        spec, raw = fo.run_batch_f32((pcm.astype(np.float32) / np.float32(32768.0))[None], want_u16=True)
        assert np.array_equal(raw[0], K[f"fe_raw/{n}"]), n                       # bit-exact integers (resolves SURVEY risks R1, R2)
        assert np.array_equal(spec[0], K[f"fe_spec/{n}"]), n                     # and the float path of input_data.py:23,34


def test_embedding_oracle_equals_keras(K):
    from multilingual_kws_amd import weights
    from oracle.efficientnet_oracle import EmbeddingOracle
    taps = {}
    EmbeddingOracle(weights.synthetic_blob(int(K["emb_weights_seed"]))).forward(K["emb_spec"][..., 0], taps)
    checked = 0
    for kname, oname in ORACLE_TAP.items():
        if f"emb_tap/{kname}" in K.files:
            ref = K[f"emb_tap/{kname}"]
            got = taps[oname].reshape(ref.shape)
            assert np.abs(got - ref).max() / np.abs(ref).max() < 1e-4, kname
            checked += 1
    assert checked >= 5 and "emb_tap/dense_2" in K.files
    assert np.array_equal(taps["dense_2"].argmax(1), K["emb_tap/dense_2"].argmax(1))


def test_head_oracle_equals_keras(K):
    from oracle import head_oracle as ho
    p0 = K["head_p0"]
    probs, _ = ho.forward(p0, K["head_emb"])
    assert probs.shape == K["head_probs"].shape                                  # (head_probs is post-update; compared below)
    loss, g, _, _ = ho.loss_and_grad(p0, K["head_emb"], K["head_labels"])
    assert abs(loss - float(K["head_loss"])) < 1e-5
    assert np.abs(g - K["head_grad"]).max() / np.abs(K["head_grad"]).max() < 1e-4
    p1 = ho.KerasAdam(len(p0), lr=1e-3).step(p0.astype(np.float64), g)
    assert np.abs(p1 - K["head_p1"]).max() < 2e-6
    probs1, _ = ho.forward(K["head_p1"], K["head_emb"])
    assert np.abs(probs1 - K["head_probs"]).max() < 1e-5


@pytest.mark.gpu
def test_device_equals_keras(K):
    import torch
    from multilingual_kws_amd import weights
    from multilingual_kws_amd.embedding_model import EmbeddingModel
    from multilingual_kws_amd.frontend import Frontend
    from multilingual_kws_amd.head import Head
    dev = torch.device("cuda:0")
    fe = Frontend()
    for n in [k.split("/", 1)[1] for k in K.files if k.startswith("fe_raw/")]:
        audio = torch.from_numpy(K[f"fe_in/{n}"].astype(np.float32) / np.float32(32768.0))[None].to(dev)
        spec, raw = fe.forward(audio, want_raw=True)
        assert np.array_equal(raw[0].cpu().numpy().view(np.uint16), K[f"fe_raw/{n}"]), n
        assert np.array_equal(spec[0].cpu().numpy(), K[f"fe_spec/{n}"]), n
    em = EmbeddingModel(weights.synthetic_blob(int(K["emb_weights_seed"])), max_batch=4)
    x = torch.from_numpy(K["emb_spec"][..., 0]).to(dev)
    for kname, oname in ORACLE_TAP.items():
        if f"emb_tap/{kname}" in K.files:
            ref = K[f"emb_tap/{kname}"]
            got = em.tap(x, oname).cpu().numpy().reshape(ref.shape)
            assert np.abs(got - ref).max() / np.abs(ref).max() < 1e-3, kname      # north_star tolerance
    emb = em.forward(x).cpu().numpy()
    assert np.array_equal(emb.argmax(1), K["emb_tap/dense_2"].argmax(1))        # label indices bit-exact
    hd = Head(params=K["head_p0"], max_batch=32)
    e, y = torch.from_numpy(K["head_emb"]).to(dev), torch.from_numpy(K["head_labels"].astype(np.int32)).to(dev)
    stats = hd.loss_grad(e, y).tolist()
    assert abs(stats[0] / 32 - float(K["head_loss"])) < 1e-4
    assert np.abs(hd.grad_view().cpu().numpy() - K["head_grad"]).max() / np.abs(K["head_grad"]).max() < 1e-3
    hd.adam_step(lr=1e-3)
    assert np.abs(hd.get_params() - K["head_p1"]).max() < 2e-6
    assert np.abs(hd.forward(e).cpu().numpy() - K["head_probs"]).max() < 1e-5
