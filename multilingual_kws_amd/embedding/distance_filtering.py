"""Embedding consumers of the hot path, same names and return contracts as the reference:

* embedding_model / cluster_and_sort      -- multilingual_kws/embedding/distance_filtering.py:12-83
* embed_files / export_keyword_embeddings -- notebooks/dataperf_experiments.py:320-338,385-415 (DataPerf export:
  one parquet per keyword with columns clip_id, mswc_embedding_vector)

Spectrograms and feature vectors come from the HIP path (one frontend launch + one embedding pass per batch
of up to 1024 clips instead of a per-file TF op); k-means stays sklearn.cluster.KMeans with the reference's
arguments, so given equal feature vectors the clustering is the reference's.
"""
import os
from pathlib import Path

import numpy as np

from . import input_data
from .transfer_learning import _specs_for_files, load_base_model


def embedding_model(base_model_path="synthetic", base_model_output="dense_2", max_batch=1024):
    """distance_filtering.py:12-28.  Returns the frozen embedding with a Keras-style .predict([N,49,40(,1)]) ->
    [N,1024].  base_model_path: weight-container directory (multilingual_kws_amd.weights.save) or "synthetic[:seed]"."""
    if base_model_output != "dense_2":
        raise NotImplementedError("the embedding is cut at dense_2 (the layer every reference call site uses)")
    model, _ = load_base_model(base_model_path, max_batch=max_batch)
    return model


def embed_files(files, embedding, model_settings=None, batch_size=1024):
    """dataperf_experiments.py:320-338: list of WAV paths -> float32 [N, 1024], in batches of `batch_size`."""
    if model_settings is None:
        model_settings = input_data.standard_microspeech_model_settings(label_count=3)
    files = [str(f) for f in files]
    out = np.zeros((len(files), 1024), dtype=np.float32)
    for s in range(0, len(files), batch_size):
        chunk = files[s:s + batch_size]
        out[s:s + len(chunk)] = embedding.predict(_specs_for_files(chunk, model_settings))
    return out


def cluster_and_sort(keyword_samples, embedding_model, seed=123, n_train=50, n_clusters=5, model_settings=None):
    """distance_filtering.py:30-83.
    Returns:
        dict(sorted_clips, cluster_centers, distances, train_clips): evaluation clips sorted by the L2 distance to
        their closest k-means centre (k-means fitted on the embeddings of n_train randomly chosen clips).
    """
    import sklearn.cluster
    if model_settings is None:
        model_settings = input_data.standard_microspeech_model_settings(label_count=761)
    assert len(keyword_samples) > n_train, f"{n_train} > number of keyword samples"

    rng = np.random.RandomState(seed)
    kwdata = rng.permutation(keyword_samples)
    train_clips = kwdata[:n_train]
    eval_clips = kwdata[n_train:]

    feature_vectors = embed_files(train_clips, embedding_model, model_settings)
    kmeans = sklearn.cluster.KMeans(n_clusters=n_clusters, random_state=seed).fit(feature_vectors)
    eval_vectors = embed_files(eval_clips, embedding_model, model_settings)

    l2_distances = np.linalg.norm(kmeans.cluster_centers_[np.newaxis].astype(np.float32) - eval_vectors[:, np.newaxis], axis=-1)
    l2_from_closest_cluster = l2_distances.min(axis=1)
    sorting = np.argsort(l2_from_closest_cluster)
    return dict(
        sorted_clips=eval_clips[sorting],
        cluster_centers=kmeans.cluster_centers_,
        distances=l2_from_closest_cluster[sorting],
        train_clips=train_clips,
    )


def export_keyword_embeddings(clips_dir, dest_dir, embedding, model_settings=None, batch_size=1024, keywords=None):
    """dataperf_experiments.py:385-415: for every keyword directory under clips_dir write
    dest_dir/<keyword>.parquet with columns clip_id (path relative to clips_dir) and mswc_embedding_vector.
    Existing parquets are skipped (resume) and so are empty keyword directories, as in the reference loop.
    Returns the list of files written."""
    import pandas as pd
    clips_dir, dest_dir = Path(clips_dir), Path(dest_dir)
    dest_dir.mkdir(parents=True, exist_ok=True)
    if keywords is None:
        keywords = list(sorted(os.listdir(clips_dir)))
    written = []
    for keyword in keywords:
        keyword_samples = list(sorted((clips_dir / keyword).glob("*.wav")))
        dest = dest_dir / f"{keyword}.parquet"
        if dest.exists() or len(keyword_samples) == 0:
            continue
        feature_vecs = embed_files(keyword_samples, embedding, model_settings, batch_size)
        id_paths = [str(fp.relative_to(clips_dir)) for fp in keyword_samples]
        df = pd.DataFrame(data=dict(clip_id=id_paths, mswc_embedding_vector=pd.Series(list(feature_vecs))))
        df.to_parquet(dest)
        written.append(dest)
    return written
