#!/opt/conda/bin/python3.9
"""Writes tests/golden/keras_h5/{weights.h5, model.h5, userblock.h5, expected.npz}: HDF5 files in the layout Keras 2.7 gives an `.h5` checkpoint
(keras/saving/hdf5_format.py: save_weights_to_hdf5_group / save_model_to_hdf5), written by h5py -- a THIRD-PARTY writer (HDF5 1.10
library) -- so that multilingual_kws_amd/checkpoint_import.py's pure-Python HDF5 reader is checked against files its author's code
did not produce.  Run with an interpreter that has h5py (this image: /opt/conda/bin/python3.9, h5py 3.3.0); the fixture is data.

Layout reproduced (reference: multilingual_kws/embedding/transfer_learning.py:36, tf.keras.models.load_model accepts `.h5`):
  weights-only file   /<layer>/<weight name>            e.g. /stem_conv/stem_conv/kernel:0
  whole-model file    /model_weights/<layer>/<weight name>, plus /optimizer_weights and the model_config / training_config attributes
  group attributes    layer_names (fixed-length byte strings), backend, keras_version; per layer group: weight_names
  nested models       one layer group named after the inner model ("efficientnetb0") holding the inner variables under their own names
  long name lists     split as weight_names0, weight_names1, ... when one attribute would exceed 64 KiB (hdf5_format.py
                      save_attributes_to_hdf5_group, HDF5_OBJECT_HEADER_LIMIT)
Tensor VALUES are small pseudo-random arrays (the full 13 M-parameter blob would be 52 MB): shapes are the real shapes divided down
so that every rank (0-d scalars, 1-d, 2-d, 4-d) and several dtypes occur.  tests/test_checkpoint_import.py also writes a full-size
file at test time when this interpreter is present.

    /opt/conda/bin/python3.9 tests/golden/make_h5_fixture.py [--full <out.h5> <blob.npy> <manifest.json>]
"""
import json
import os
import sys

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "keras_h5")
HDF5_OBJECT_HEADER_LIMIT = 64512


def save_attributes(group, name, data):
    """keras/saving/hdf5_format.py save_attributes_to_hdf5_group, restated: one attribute, or name0, name1, ... chunks."""
    bad = [x for x in data if len(x) > HDF5_OBJECT_HEADER_LIMIT]
    assert not bad
    arr = np.asarray(data)
    n = 1
    chunks = np.array_split(arr, n)
    while any(x.nbytes > HDF5_OBJECT_HEADER_LIMIT for x in chunks):
        n += 1
        chunks = np.array_split(arr, n)
    if n > 1:
        for i, c in enumerate(chunks):
            group.attrs["%s%d" % (name, i)] = c
    else:
        group.attrs[name] = data


def save_weights(f, layers):
    """layers: [(layer name, [(weight name, array)])] -> Keras' group layout under f."""
    save_attributes(f, "layer_names", [n.encode("utf8") for n, _ in layers])
    f.attrs["backend"] = "tensorflow".encode("utf8")
    f.attrs["keras_version"] = "2.7.0".encode("utf8")
    for lname, ws in layers:
        g = f.create_group(lname)
        names = [w.encode("utf8") for w, _ in ws]
        if lname == "dense_2":        # older h5py / explicit numpy input: FIXED-length strings instead of the variable-length ones h5py 3 picks for a list
            g.attrs["weight_names"] = np.asarray(names)
        elif lname == "dense_1":      # the chunked form Keras falls back to above 64 KiB per attribute (forced here: the reader must join name0, name1, ...)
            for i, c in enumerate(np.array_split(np.asarray(names), 2)):
                g.attrs["weight_names%d" % i] = c
        else:
            save_attributes(g, "weight_names", names)
        for wname, val in ws:
            d = g.create_dataset(wname, val.shape, dtype=val.dtype)
            if not val.shape:
                d[()] = val
            else:
                d[:] = val


def layers_from_named(named, order, nested="efficientnetb0"):
    """The reference model = [inner EfficientNetB0 model, global pooling, dense, dense_1, dense_2] (train_multilingual_embedding.py:58-83):
    trunk variables live in ONE layer group named after the inner model, the dense layers in groups of their own."""
    trunk, dense = [], {}
    for name in order:
        layer = name.split("/")[0]
        if layer.startswith("dense"):
            dense.setdefault(layer, []).append((name + ":0", named[name]))
        else:
            trunk.append((name + ":0", named[name]))
    return [(nested, trunk), ("global_average_pooling2d", [])] + [(k, v) for k, v in dense.items()]


def main():
    if len(sys.argv) >= 5 and sys.argv[1] == "--full":
        out, blob, manifest = sys.argv[2], np.load(sys.argv[3]), json.load(open(sys.argv[4]))["tensors"]
        named = {t["name"]: blob[t["offset"]:t["offset"] + t["count"]].reshape(t["shape"]).astype(np.float32) for t in manifest}
        with h5py.File(out, "w") as f:
            save_weights(f.create_group("model_weights"), layers_from_named(named, [t["name"] for t in manifest]))
            f.attrs["model_config"] = json.dumps({"class_name": "Functional"}).encode("utf8")
        return
    os.makedirs(OUT, exist_ok=True)
    manifest = json.load(open(os.path.join(os.path.dirname(os.path.dirname(HERE)), "tools", "embedding_manifest.json")))["tensors"]
    rng = np.random.default_rng(20260927)
    named, order = {}, []
    for k, t in enumerate(manifest):
        shape = [max(1, d // 16) if d > 16 else d for d in t["shape"]]               # real ranks, reduced extents
        named[t["name"]] = (rng.standard_normal(shape) * (1 + k % 5)).astype(np.float32)
        order.append(t["name"])
    named["normalization/count"] = np.asarray(rng.integers(1, 1 << 40), np.int64); order.insert(2, "normalization/count")      # Keras' third Normalization variable
    # extras that exercise the reader: a float64 tensor, a 0-d dataset, an empty layer, a big-endian dataset
    named["dense_2/extra_f64"] = rng.standard_normal((3, 5)); order.append("dense_2/extra_f64")
    named["dense_2/scalar"] = np.float32(2.5); order.append("dense_2/scalar")
    named["dense_2/big_endian"] = rng.standard_normal((4, 3)).astype(">f4"); order.append("dense_2/big_endian")
    layers = layers_from_named(named, order)
    # the same weights file behind a 512-byte USER BLOCK (superblock at 512, base address 512; libhdf5 stores the end-of-file
    # address as an ABSOLUTE offset, object addresses relative to the base): written by h5py, not patched together
    with h5py.File(os.path.join(OUT, "userblock.h5"), "w", userblock_size=512, libver="earliest") as f:
        save_weights(f, layers)
    with open(os.path.join(OUT, "userblock.h5"), "r+b") as fh:
        fh.write(b"#!user block: 512 bytes the HDF5 library never reads\n")
    if "--userblock-only" in sys.argv:
        print("wrote", os.path.join(OUT, "userblock.h5"), os.path.getsize(os.path.join(OUT, "userblock.h5")))
        return
    with h5py.File(os.path.join(OUT, "weights.h5"), "w") as f:                     # model.save_weights("x.h5")
        save_weights(f, layers)
    with h5py.File(os.path.join(OUT, "model.h5"), "w") as f:                       # model.save("x.h5")
        f.attrs["keras_version"] = "2.7.0".encode("utf8")
        f.attrs["backend"] = "tensorflow".encode("utf8")
        f.attrs["model_config"] = json.dumps({"class_name": "Functional", "config": {"name": "model", "layers": ["..."] * 50}}).encode("utf8")
        f.attrs["training_config"] = json.dumps({"loss": "sparse_categorical_crossentropy"}).encode("utf8")
        save_weights(f.create_group("model_weights"), layers)
        og = f.create_group("optimizer_weights")
        save_attributes(og, "weight_names", [b"Adam/iter:0"])
        og.create_dataset("Adam/iter:0", (), dtype=np.int64)[()] = 12345
        # a chunked + compressed dataset: the reader must refuse it loudly, not misread it
        f.create_dataset("compressed_extra", data=rng.standard_normal((64, 64)).astype(np.float32), chunks=(16, 16), compression="gzip")
    np.savez_compressed(os.path.join(OUT, "expected.npz"), **{k.replace("/", "|"): np.asarray(v, dtype=np.asarray(v).dtype.newbyteorder("=")) for k, v in named.items()})
    print("wrote", OUT, {n: os.path.getsize(os.path.join(OUT, n)) for n in sorted(os.listdir(OUT))}, "h5py", h5py.__version__, "HDF5", h5py.version.hdf5_version)


if __name__ == "__main__":
    main()
