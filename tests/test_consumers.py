"""Embedding consumers: distance_filtering.cluster_and_sort and the DataPerf parquet
export.  CPU tests stub the spectrogram/embedding steps (host logic and return contracts); the GPU test runs
the real hot path on synthetic WAV files."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(__file__))
import util_data  # noqa: E402


class _StubEmbedding:
    """Deterministic stand-in: feature = fixed random projection of the spectrogram mean/std per channel."""
    def __init__(self):
        self.P = np.random.default_rng(5).standard_normal((80, 1024)).astype(np.float32)

    def predict(self, specs):
        specs = np.asarray(specs, dtype=np.float32).reshape(len(specs), 49, 40)
        f = np.concatenate([specs.mean(1), specs.std(1)], axis=1)
        return f @ self.P


def _stub_specs(files, model_settings):
    # one pseudo-spectrogram per file name: 3 families (by the digit in the name) with small jitter
    out = []
    for f in files:
        k = int("".join(ch for ch in os.path.basename(str(f)) if ch.isdigit()))
        rng = np.random.default_rng(k)
        base = np.full((49, 40), 5.0 + 7.0 * (k % 3), np.float32)
        out.append(base + 0.3 * rng.standard_normal((49, 40)).astype(np.float32))
    return np.stack(out) if out else np.zeros((0, 49, 40), np.float32)


def test_cluster_and_sort_contract(monkeypatch):
    from multilingual_kws_amd.embedding import distance_filtering as dfl
    monkeypatch.setattr(dfl, "_specs_for_files", _stub_specs)
    files = np.array([f"/data/kw/clip{i}.wav" for i in range(80)])
    emb = _StubEmbedding()
    r = dfl.cluster_and_sort(files, emb, seed=123, n_train=50, n_clusters=5)
    assert set(r) == {"sorted_clips", "cluster_centers", "distances", "train_clips"}
    assert r["cluster_centers"].shape == (5, 1024) and len(r["train_clips"]) == 50 and len(r["sorted_clips"]) == 30
    assert np.all(np.diff(r["distances"]) >= 0)                                   # sorted by distance
    assert set(r["sorted_clips"]) | set(r["train_clips"]) == set(files)             # a permutation split
    # the split is numpy's RandomState(seed).permutation, as in the reference
    perm = np.random.RandomState(123).permutation(files)
    assert list(r["train_clips"]) == list(perm[:50])
    # distances really are the min L2 to the returned centres
    ev = emb.predict(_stub_specs(r["sorted_clips"], None))
    d = np.linalg.norm(r["cluster_centers"][None].astype(np.float32) - ev[:, None], axis=-1).min(1)
    assert np.allclose(d, r["distances"], rtol=1e-5)
    r2 = dfl.cluster_and_sort(files, emb, seed=123, n_train=50, n_clusters=5)
    assert list(r2["sorted_clips"]) == list(r["sorted_clips"])                      # deterministic
    with pytest.raises(AssertionError):
        dfl.cluster_and_sort(files[:50], emb, n_train=50)


def test_export_keyword_embeddings_layout(monkeypatch, tmp_path):
    pytest.importorskip("pyarrow")
    import pandas as pd
    from multilingual_kws_amd.embedding import distance_filtering as dfl
    monkeypatch.setattr(dfl, "_specs_for_files", _stub_specs)
    clips = tmp_path / "clips"
    for kw, n in (("alpha", 3), ("beta", 2), ("empty", 0)):
        (clips / kw).mkdir(parents=True)
        for i in range(n):
            (clips / kw / f"c{i}.wav").write_bytes(b"")                            # never decoded (stubbed)
    dest = tmp_path / "emb"
    written = dfl.export_keyword_embeddings(clips, dest, _StubEmbedding(), batch_size=2)
    assert sorted(p.name for p in written) == ["alpha.parquet", "beta.parquet"]      # empty keyword skipped
    df = pd.read_parquet(dest / "alpha.parquet")
    assert list(df.columns) == ["clip_id", "mswc_embedding_vector"]
    assert list(df["clip_id"]) == ["alpha/c0.wav", "alpha/c1.wav", "alpha/c2.wav"]
    assert len(df["mswc_embedding_vector"][0]) == 1024
    assert dfl.export_keyword_embeddings(clips, dest, _StubEmbedding()) == []       # resume: nothing rewritten


@pytest.mark.gpu
def test_consumers_on_device(tmp_path):
    import pandas as pd
    from multilingual_kws_amd.embedding import distance_filtering as dfl, input_data
    rng = np.random.default_rng(3)
    files = []
    for i in range(24):
        p = str(tmp_path / "clips" / "kw" / f"k{i:02d}.wav")
        util_data.write_wav(p, util_data.tone_clip(500 + 400 * (i % 3), rng, burst=(2000, 12000)))
        files.append(p)
    emb = dfl.embedding_model("synthetic")
    vec = dfl.embed_files(files, emb, batch_size=10)                                # ragged batches
    assert vec.shape == (24, 1024) and np.isfinite(vec).all()
    settings = input_data.standard_microspeech_model_settings(3)
    one = emb.predict(input_data.file2spec(settings, files[7])[None])
    assert np.allclose(vec[7], one[0], rtol=1e-5, atol=1e-6)                        # batching does not change a clip's vector
    # the vectors themselves, against the oracle chain (WAV decode on the host, C micro-frontend, fp32 EfficientNet-B0)
    from multilingual_kws_amd import weights
    from oracle.efficientnet_oracle import EmbeddingOracle
    from oracle.frontend_oracle import FrontendOracle
    pick = [0, 7, 23]
    audio = np.stack([input_data.decode_wav(open(files[i], "rb").read(), 16000)[0] for i in pick]).astype(np.float32)
    ref_spec = FrontendOracle().run_batch_f32(audio)
    assert np.array_equal(np.stack([input_data.file2spec(settings, files[i]) for i in pick]), ref_spec)      # frontend: bit-exact
    ref = EmbeddingOracle(weights.synthetic_blob()).forward(ref_spec).numpy()
    assert np.abs(vec[pick] - ref).max() / np.abs(ref).max() < 1e-4
    r = dfl.cluster_and_sort(np.array(files), emb, seed=1, n_train=15, n_clusters=3)
    assert len(r["sorted_clips"]) == 9 and np.all(np.diff(r["distances"]) >= 0)
    try:
        import pyarrow  # noqa: F401  (absent on some GPU boxes: the parquet layout itself is covered by the CPU test)
    except ImportError:
        return
    written = dfl.export_keyword_embeddings(tmp_path / "clips", tmp_path / "out", emb)
    df = pd.read_parquet(written[0])
    assert len(df) == 24 and np.allclose(np.stack(df["mswc_embedding_vector"].to_numpy()), vec, rtol=1e-6)
