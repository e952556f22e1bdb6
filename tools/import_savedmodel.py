#!/usr/bin/env python
"""Keras SavedModel directory (e.g. the reference's multilingual_context_73_0.8011) -> this repo's weight container,
WITHOUT TensorFlow (multilingual_kws_amd/checkpoint_import.py reads the variables bundle directly).

    python tools/import_savedmodel.py /path/to/multilingual_context_73_0.8011 out_dir

(transfer_learn / load_base_model also accept the SavedModel directory itself.)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == "__main__":
    from multilingual_kws_amd import checkpoint_import, weights
    src, dst = sys.argv[1], sys.argv[2]
    blob = checkpoint_import.import_savedmodel(src)
    weights.save(dst, blob)
    print(f"wrote {blob.shape[0]} floats to {dst}")
