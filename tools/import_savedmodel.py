#!/usr/bin/env python
"""Keras SavedModel directory (e.g. the reference's multilingual_context_73_0.8011) -> this repo's weight container,
WITHOUT TensorFlow (multilingual_kws_amd/checkpoint_import.py reads the variables bundle directly).

    python tools/import_savedmodel.py /path/to/multilingual_context_73_0.8011 out_dir
    python tools/import_savedmodel.py --verify /path/to/multilingual_context_73_0.8011

    python tools/import_savedmodel.py /path/to/model.h5 out_dir            (a Keras .h5 checkpoint: pure-Python HDF5 reader)

(transfer_learn / load_base_model also accept the SavedModel directory or the .h5 file itself.)

--verify imports nothing: it walks the variables bundle (<dir>/variables/variables.* or <dir>/variables.*), checks every
block CRC of the index and every tensor's stored CRC-32C against its bytes, and prints one line per tensor

    <checkpoint key> <dtype> <shape> <bytes> crc32c ok|MISMATCH|absent sha1 <sha1 of the tensor bytes> [-> <Keras variable name>]

followed by whether the embedding architecture's 300+ tensors are all present -- so that whoever holds the released
checkpoint (docker/Dockerfile:69-70 of the reference) can confirm the reader against the real file in one command and
compare the sha1s with `tf.train.load_checkpoint(...).get_tensor(key).tobytes()`.  Exit code 1 on any mismatch."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def verify_h5(path):
    """One line per dataset of a Keras .h5 file: <variable> <dtype> <shape> <bytes> sha1 (compare with h5py: hashlib.sha1(np.asarray(d).tobytes()))."""
    import numpy as np
    from multilingual_kws_amd import checkpoint_import as ci, weights
    r = ci.load_h5(path)
    for name, v in r["named"].items():
        v = np.ascontiguousarray(v)
        print(f"{name} {v.dtype.name} {list(v.shape)} {v.nbytes} sha1 {hashlib.sha1(v.tobytes()).hexdigest()}")
    missing = [t["name"] for t in weights.manifest() if t["name"] not in r["named"]]
    print(f"# {len(r['named'])} variables in layers {[l for l, _ in r['layers']]}; embedding architecture: {len(weights.manifest()) - len(missing)} of "
          f"{len(weights.manifest())} tensors found by name" + (f"; missing e.g. {missing[:3]}" if missing else ""))
    return 1 if missing else 0


def verify(path):
    import numpy as np
    from multilingual_kws_amd import checkpoint_import as ci
    if os.path.isfile(path):
        return verify_h5(path)
    prefix = os.path.join(path, "variables", "variables")
    if not os.path.exists(prefix + ".index"):
        prefix = os.path.join(path, "variables")
    if not os.path.exists(prefix + ".index"):
        raise SystemExit(f"{path}: no variables/variables.index (SavedModel) and no variables.index (bare bundle)")
    rd = ci.BundleReader(prefix, verify=True)                  # raises on a bad index block
    names = {}
    try:
        info = ci.load_savedmodel(path, verify=True) if os.path.exists(os.path.join(path, "variables", "variables.index")) else None
        if info:
            names = {v: k for k, v in info["named"].items()}
    except Exception as exc:                                    # a bare bundle has no object graph: keys only
        print(f"# no variable names: {exc}")
    bad = n = 0
    for key in sorted(rd.entries):
        e = rd.entries[key]
        if e["dtype"] == ci.DT_STRING:
            print(f"{key} string {e['shape']} {e['size']} (not checked)")
            continue
        raw = bytes(rd.raw(key))
        if e["crc32c"] is None:
            verdict = "absent"
        elif ci.mask_crc(ci.crc32c(raw)) == e["crc32c"]:
            verdict = "ok"
        else:
            verdict, bad = "MISMATCH", bad + 1
        n += 1
        dt = ci.DTYPES.get(e["dtype"])
        print(f"{key} {np.dtype(dt).name if dt is not None else 'dtype' + str(e['dtype'])} {e['shape']} {len(raw)} crc32c {verdict} "
              f"sha1 {hashlib.sha1(raw).hexdigest()}" + (f" -> {names[key]}" if key in names else ""))
    print(f"# {n} tensors, {bad} bad")
    if names:
        from multilingual_kws_amd import weights
        missing = [t["name"] for t in weights.manifest() if t["name"] not in set(names.values())]
        print(f"# embedding architecture: {len(weights.manifest()) - len(missing)} of {len(weights.manifest())} tensors found by name"
              + (f"; missing e.g. {missing[:3]} (the importer then matches layers positionally)" if missing else ""))
    return 1 if bad else 0


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--verify":
        raise SystemExit(verify(sys.argv[2]))
    if len(sys.argv) != 3:
        raise SystemExit(__doc__)
    from multilingual_kws_amd import checkpoint_import, weights
    src, dst = sys.argv[1], sys.argv[2]
    blob = checkpoint_import.import_h5(src) if os.path.isfile(src) else checkpoint_import.import_savedmodel(src)      # a Keras .h5 file, or a SavedModel directory
    weights.save(dst, blob)
    print(f"wrote {blob.shape[0]} floats to {dst}")
