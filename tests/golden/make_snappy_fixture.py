#!/usr/bin/env python3
"""Writes tests/golden/snappy_vectors.npz: byte strings compressed by the THIRD-PARTY snappy library of this image
(/opt/conda/lib/libsnappy.so.1.1.8, Google's reference implementation, through its C API snappy-c.h via ctypes), for
multilingual_kws_amd/checkpoint_import.py's snappy_decompress -- the decoder a TensorFlow `variables.index` SSTable needs
(tensorflow/core/lib/io/format.cc, block compression type 1; reference: multilingual_kws/embedding/transfer_learning.py:36 loads such a
checkpoint).  Until this fixture the decoder had only hand-made known-answer vectors by its own author.  The fixture is data:
(input, compressed) pairs; the test needs no snappy library.

Inputs are chosen to make the encoder emit every element type the format has: literals of all length classes (inline, 1-, 2-, 3-byte
lengths), copies with 1-byte offsets (tag 01, lengths 4-11, offsets < 2048), 2-byte offsets (tag 10), long matches split into 64-byte
copies, overlapping copies (run-length: offset < length), incompressible data, the empty string, and SSTable-like blocks (prefix-compressed
checkpoint keys + protobuf values)."""
import ctypes
import os

import numpy as np

LIB = "/opt/conda/lib/libsnappy.so.1.1.8"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "snappy_vectors.npz")


def main():
    L = ctypes.CDLL(LIB)
    L.snappy_max_compressed_length.restype = ctypes.c_size_t
    L.snappy_max_compressed_length.argtypes = [ctypes.c_size_t]
    L.snappy_compress.restype = ctypes.c_int
    L.snappy_compress.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.POINTER(ctypes.c_size_t)]
    L.snappy_uncompress.restype = ctypes.c_int
    L.snappy_uncompress.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.POINTER(ctypes.c_size_t)]

    def compress(b):
        n = ctypes.c_size_t(L.snappy_max_compressed_length(len(b)))
        buf = ctypes.create_string_buffer(n.value)
        assert L.snappy_compress(b, len(b), buf, ctypes.byref(n)) == 0
        out = buf.raw[:n.value]
        m = ctypes.c_size_t(len(b) + 1)
        back = ctypes.create_string_buffer(m.value)
        assert L.snappy_uncompress(out, len(out), back, ctypes.byref(m)) == 0 and back.raw[:m.value] == b
        return out

    rng = np.random.default_rng(7)
    keys = [f"layer_with_weights-{i}/{leaf}/.ATTRIBUTES/VARIABLE_VALUE".encode() for i in range(120) for leaf in ("kernel", "bias", "gamma", "beta", "moving_mean")]
    sst = b"".join(bytes([min(len(k), 40), len(k), 24]) + k + rng.integers(0, 256, 24, dtype=np.uint8).tobytes() for k in keys)
    cases = {
        "empty": b"",
        "one_byte": b"x",
        "short_literal": b"hello, snappy",
        "literal_60": bytes(range(60)),
        "literal_61": bytes(range(61)),                                            # first 1-byte literal length
        "literal_300": rng.integers(0, 256, 300, dtype=np.uint8).tobytes(),          # 2-byte literal length
        "random_70000": rng.integers(0, 256, 70000, dtype=np.uint8).tobytes(),       # 3-byte literal length, two 64 KiB blocks, incompressible
        "run_length": b"a" * 5000,                                                   # overlapping copies (offset 1)
        "period_7": b"abcdefg" * 3000,
        "text": (b"The quick brown fox jumps over the lazy dog. " * 400) + b"Pack my box with five dozen liquor jugs. " * 300,
        "far_copies": b"".join(rng.integers(0, 256, 64, dtype=np.uint8).tobytes() for _ in range(40)) * 30,        # matches thousands of bytes back: 2-byte offsets
        "floats": np.repeat(rng.standard_normal(600).astype(np.float32), 5).tobytes(),
        "zeros_then_noise": b"\x00" * 40000 + rng.integers(0, 256, 3000, dtype=np.uint8).tobytes() + b"\x00" * 30000,
        "sstable_like": sst,
    }
    out = {}
    tags = np.zeros(4, np.int64)
    for name, b in cases.items():
        c = compress(b)
        out["in/" + name] = np.frombuffer(b, np.uint8)
        out["z/" + name] = np.frombuffer(c, np.uint8)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", {k: (len(v), len(compress(v))) for k, v in cases.items()})


if __name__ == "__main__":
    main()
