"""-m gpu: row f2 ON THE DEVICE -- an on-disk Keras checkpoint goes through the reference's own call
(`tf.keras.models.load_model(base_model_path)` + cut at `dense_2`, reference transfer_learning.py:36-43 =
`transfer_learning.load_base_model(path)` here, UNPATCHED) into the HIP embedding, and the result is held to the
oracle built from the tensors that were written.

Two formats, both at the full 13 M-parameter size, both written at test time:
  * SavedModel directory (`variables/variables.index` + data shards: SSTable + tensor bundle + object graph) -- written by
    tests/util_bundle.py, this project's own writer from the format specs.  NO TensorFlow-written SavedModel has ever been read
    (INTEGRATION.md says so next to the claim that `multilingual_context_73_0.8011` loads).
  * Keras whole-model `.h5` -- written by h5py + libhdf5 (a third-party writer; /opt/conda/bin/python3.9 in this image).
"""
import os
import subprocess

import numpy as np
import pytest

from tests.test_checkpoint_import import H5PY_PYTHON, _write_model

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _blob(k):
    """The calibrated synthetic weights (well conditioned: every layer input-dependent), made DIFFERENT from what "synthetic"
    loads so that a loader silently falling back to defaults could not pass: a few tensors rescaled / shifted by k."""
    from multilingual_kws_amd import weights
    blob = weights.synthetic_blob().copy()
    for t in weights.manifest():
        sl = slice(t["offset"], t["offset"] + t["count"])
        if t["name"] in ("stem_conv/kernel", "top_conv/kernel", "dense_1/kernel"):
            blob[sl] *= np.float32(1.0 + 0.02 * k)
        elif t["name"] in ("dense_2/bias", "block4a_se_reduce/bias"):
            blob[sl] += np.float32(0.01 * k)
    assert not np.array_equal(blob, weights.synthetic_blob())
    return blob


def _specs(n, seed):
    """Features of real frontend runs would do; what matters here is the value grid k * 10/256 the network was calibrated on."""
    rng = np.random.default_rng(seed)
    s = rng.integers(0, 670, size=(n, 49, 40)).astype(np.float32) * np.float32(10 / 256)
    s[1, 30:] = 0.0                                   # a clip whose tail is silent (zero-padded audio)
    return s


def _check_against_oracle(emb, blob_written, blob_returned, seed):
    from oracle import head_oracle as ho
    from oracle.efficientnet_oracle import EmbeddingOracle
    from multilingual_kws_amd.head import Head, glorot_uniform_params
    assert blob_returned.dtype == np.float32 and np.array_equal(blob_returned, blob_written)       # the importer is bit-exact
    spec = _specs(8, seed)
    ref = EmbeddingOracle(blob_written).forward(spec).numpy()
    got = emb.forward(torch.from_numpy(spec).to("cuda:0")).cpu().numpy()
    assert got.shape == (8, 1024)
    assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max()                                       # measured ~1e-6
    assert (np.abs(got - ref) <= 1e-3 * np.abs(ref) + 1e-5 * np.abs(ref).max()).all()                # north_star's bound, element by element
    assert np.array_equal(got.argmax(1), ref.argmax(1))
    # ... and through a few-shot head: class probabilities and their argmax (the labels the reference's callers read)
    params = glorot_uniform_params(1024, 18, 3, seed)
    head = Head(1024, 18, 3, max_batch=8, params=params)
    probs = head.forward(emb.forward(torch.from_numpy(spec).to("cuda:0"))).cpu().numpy()
    ref_probs, _ = ho.forward(params, ref)
    assert np.abs(probs - ref_probs).max() < 1e-4 and np.array_equal(probs.argmax(1), ref_probs.argmax(1))


def test_savedmodel_directory_through_the_hip_path(tmp_path):
    from multilingual_kws_amd import weights
    from multilingual_kws_amd.embedding import transfer_learning as tl
    blob = _blob(1)
    path = _write_model(tmp_path, blob, with_full_names=True, name_prefix="efficientnetb0/", compress=True, block_entries=16, num_shards=2)
    assert os.path.exists(os.path.join(path, "variables", "variables.index"))
    emb, got_blob = tl.load_base_model(path, max_batch=8)                  # base_model_output defaults to "dense_2"
    _check_against_oracle(emb, blob, got_blob, seed=1)


def test_savedmodel_without_variable_names_through_the_hip_path(tmp_path):
    """Checkpoints whose object graph carries no `full_name`s are matched positionally (Keras layer order)."""
    from multilingual_kws_amd import weights
    from multilingual_kws_amd.embedding import transfer_learning as tl
    blob = _blob(2)
    path = _write_model(tmp_path, blob, with_full_names=False)
    emb, got_blob = tl.load_base_model(path, max_batch=8)
    _check_against_oracle(emb, blob, got_blob, seed=2)


@pytest.mark.skipif(not os.path.exists(H5PY_PYTHON), reason="no interpreter with h5py on this box")
def test_keras_h5_written_by_h5py_through_the_hip_path(tmp_path, golden_dir):
    from multilingual_kws_amd import weights
    from multilingual_kws_amd.embedding import transfer_learning as tl
    blob = _blob(3)
    np.save(tmp_path / "blob.npy", blob)
    manifest = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "embedding_manifest.json")
    out = str(tmp_path / "multilingual_context_73_0.8011.h5")
    subprocess.run([H5PY_PYTHON, os.path.join(golden_dir, "make_h5_fixture.py"), "--full", out, str(tmp_path / "blob.npy"), manifest], check=True,
                   env={k: v for k, v in os.environ.items() if not k.startswith("PYTHON")})
    emb, got_blob = tl.load_base_model(out, max_batch=8)
    _check_against_oracle(emb, blob, got_blob, seed=3)


def test_transfer_learn_on_a_savedmodel_path(tmp_path):
    """The reference's canonical call with `base_model_path` = a SavedModel directory, as its users pass it (run.py:281-298); the
    returned model must predict what the oracle chain predicts from the tensors in that directory."""
    from multilingual_kws_amd import weights
    from multilingual_kws_amd.embedding import input_data, transfer_learning as tl
    from oracle import head_oracle as ho
    from oracle.efficientnet_oracle import EmbeddingOracle
    from tests.util_data import make_fewshot_dataset
    data = make_fewshot_dataset(str(tmp_path / "fewshot"))
    blob = _blob(4)
    path = _write_model(tmp_path, blob, with_full_names=True, name_prefix="efficientnetb0/")
    ms = input_data.standard_microspeech_model_settings(3)
    name, model, details = tl.transfer_learn(
        target="target", train_files=data["train"], val_files=data["val"], unknown_files=data["unknown"],
        num_epochs=1, num_batches=4, batch_size=32, primary_lr=0.001, backprop_into_embedding=False, embedding_lr=0,
        model_settings=ms, base_model_path=path, base_model_output="dense_2", bg_datadir=data["bg_dir"], verbose=0, seed=3)
    specs = np.stack([input_data.file2spec(ms, f) for f in data["val"]])
    preds = model.predict(specs[..., None])
    ref_probs, _ = ho.forward(model.head.get_params(), EmbeddingOracle(blob).forward(specs).numpy())
    assert np.abs(preds - ref_probs).max() < 1e-4 and np.array_equal(preds.argmax(1), ref_probs.argmax(1))
    model.save(str(tmp_path / "m"))                    # the saved model carries its own copy of the base weights
    again = tl.TransferLearnedModel.load(str(tmp_path / "m"), max_batch=model.embedding.max_batch)
    assert np.array_equal(again.predict(specs[..., None]), preds)
