#!/usr/bin/env python
"""Converts the reference's trained embedding checkpoint (the Keras SavedModel directory
`multilingual_context_73_0.8011`, docker/Dockerfile:69-70) into this repo's weight container.

RUN THIS WHERE TENSORFLOW IS INSTALLED (it is not in the MI355X image, so this script is untested here):

    python tools/export_keras_checkpoint.py /path/to/multilingual_context_73_0.8011 out_dir [--check]

It only needs tensorflow + numpy on that machine and this repo's `multilingual_kws_amd/weights.py` for the
tensor order.  The container (out_dir/weights.bin + manifest.json) then loads anywhere with
`transfer_learning.load_base_model(out_dir)` / `mkws_embed_create`.

Name mapping: Keras variable names minus the ":0" suffix (nested-model prefixes dropped), e.g.
`stem_conv/kernel`, `block2a_expand_bn/moving_mean`, `normalization/variance`, `dense_2/bias` -- exactly the
names `mkws_embed_weight_manifest` lists.  `--check` also writes `check_vectors.npz` (8 random spectrogram-like
inputs and the TF model's dense_2 outputs) to compare with `EmbeddingModel.predict` on the GPU box.
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def expected_tensors():
    """[{name, shape, offset, count}] -- from the C library when it is built, else from its committed manifest dump."""
    try:
        from multilingual_kws_amd import weights
        return weights.manifest()
    except Exception:
        path = os.path.join(os.path.dirname(__file__), "embedding_manifest.json")
        return json.load(open(path))["tensors"]


def collect(model):
    named = {}

    def walk(layer):
        for v in getattr(layer, "weights", []):
            name = v.name.split(":")[0]
            parts = name.split("/")
            named["/".join(parts[-2:])] = v.numpy()      # drop nested-model prefixes ("efficientnetb0/...")
        for sub in getattr(layer, "layers", []):
            walk(sub)
    walk(model)
    return named


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("saved_model_dir")
    ap.add_argument("out_dir")
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    import tensorflow as tf
    model = tf.keras.models.load_model(args.saved_model_dir)
    named = collect(model)
    tensors = expected_tensors()
    n = tensors[-1]["offset"] + tensors[-1]["count"]
    blob = np.zeros(n, dtype="<f4")
    for t in tensors:
        if t["name"] not in named:
            raise SystemExit(f"checkpoint has no variable {t['name']} (found e.g. {sorted(named)[:5]})")
        v = np.asarray(named[t["name"]], dtype=np.float32)
        if tuple(v.shape) != tuple(t["shape"]):
            raise SystemExit(f"{t['name']}: checkpoint shape {v.shape} != expected {tuple(t['shape'])}")
        blob[t["offset"]:t["offset"] + t["count"]] = v.reshape(-1)
    os.makedirs(args.out_dir, exist_ok=True)
    blob.tofile(os.path.join(args.out_dir, "weights.bin"))
    json.dump({"format": "mkws-embedding-v1", "dtype": "float32", "tensors": tensors},
              open(os.path.join(args.out_dir, "manifest.json"), "w"))
    print(f"wrote {n} floats ({len(tensors)} tensors) to {args.out_dir}")
    if args.check:
        emb = tf.keras.models.Model(inputs=model.inputs, outputs=model.get_layer(name="dense_2").output)
        rng = np.random.default_rng(0)
        x = (rng.integers(0, 670, size=(8, 49, 40, 1)).astype(np.float32) * np.float32(10 / 256))
        np.savez(os.path.join(args.out_dir, "check_vectors.npz"), spectrograms=x, dense_2=emb.predict(x))
        print("wrote check_vectors.npz")


if __name__ == "__main__":
    main()
