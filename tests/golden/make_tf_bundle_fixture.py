#!/usr/bin/env python3
"""Writes tests/golden/tf_bundle/{variables.index, variables.data-00000-of-00001, expected.json}: a tiny TensorFlow tensor
bundle assembled BYTE BY BYTE from the format specifications, to test multilingual_kws_amd/checkpoint_import.py against
something that is neither its own code nor the writer of tests/util_bundle.py.

This script shares NOTHING with either (own CRC-32C, own varints, the snappy stream is written out element by element), and
every byte below names the rule of the specification it follows:

  [T]  LevelDB table format  (leveldb/doc/table_format.md; TensorFlow's port: tensorflow/core/lib/io/{format,block_builder,
       table_builder}.cc -- same layout, same magic)
  [B]  tensorflow/core/protobuf/tensor_bundle.proto (BundleHeaderProto, BundleEntryProto) and
       tensorflow/core/util/tensor_bundle/tensor_bundle.cc (key "" = header; data file name; per-tensor crc32c)
  [P]  protobuf wire format (tag = field << 3 | wire type; 0 varint, 2 length-delimited, 5 fixed32)
  [S]  snappy format_description.txt (preamble varint = uncompressed length; element tag low 2 bits: 00 literal, 01 copy with
       1-byte offset, 10 copy with 2-byte offset)
  [C]  CRC-32C (Castagnoli, reflected polynomial 0x82F63B78, init/xorout 0xFFFFFFFF) and LevelDB's / TensorFlow's mask:
       ((crc >> 15) | (crc << 17)) + 0xa282ead8   (leveldb/util/crc32c.h, tensorflow/core/lib/hash/crc32c.h)

Contents: two float32 tensors,
   "a/bias/.ATTRIBUTES/VARIABLE_VALUE"    shape [3]
   "a/kernel/.ATTRIBUTES/VARIABLE_VALUE"  shape [2,3]
in ONE data block holding three entries (header + 2 tensors) with key prefix compression and a restart array of two
restart points, stored SNAPPY-compressed (one literal, one 2-byte-offset copy, one literal); an empty metaindex block, an
index block, the 48-byte footer.  (TensorFlow itself writes bundle indexes uncompressed; compression type 1 is legal in the
format and the reader must handle it.)
"""
import hashlib
import json
import os
import struct

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "tf_bundle")


# ---- [C] -----------------------------------------------------------------------------------------------------------
def crc32c(data):
    c = 0xFFFFFFFF
    for b in data:
        c ^= b
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
    return c ^ 0xFFFFFFFF


def masked(c):
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


assert crc32c(b"123456789") == 0xE3069283          # the standard CRC-32C check value


# ---- [P] -----------------------------------------------------------------------------------------------------------
def varint(n):
    out = bytearray()
    while n >= 0x80:
        out.append((n & 0x7F) | 0x80)
        n >>= 7
    out.append(n)
    return bytes(out)


def main():
    # ================= data file: raw little-endian tensor bytes, back to back [B] =====================================
    bias = struct.pack("<3f", 0.5, -1.25, 2.0)                                   # 12 bytes at offset 0
    kernel = struct.pack("<6f", 3.5, -0.0, 100.0, 1e-3, -7.0, 0.333251953125)    # 24 bytes at offset 12
    data_file = bias + kernel

    # ================= values of the three table entries ==============================================================
    # key "": BundleHeaderProto  [B][P]
    header = bytes([
        0x08, 0x01,              # field 1 (num_shards), varint: 1
        # field 2 (endianness) omitted: default LITTLE = 0
        0x1A, 0x02,              # field 3 (version: VersionDef), length-delimited, 2 bytes
        0x08, 0x01,              #   VersionDef.producer (field 1) = 1   (kTensorBundleVersion)
    ])

    def entry(shape, offset, size, raw):                                          # BundleEntryProto [B][P]
        dims = b"".join(bytes([0x12, 0x02, 0x08, d]) for d in shape)             # TensorShapeProto.dim (field 2) { size (field 1) = d }
        return (bytes([0x08, 0x01])                                               # field 1 dtype = DT_FLOAT (1)
                + bytes([0x12, len(dims)]) + dims                                 # field 2 shape
                # field 3 shard_id = 0: default, omitted
                + (bytes([0x20]) + varint(offset) if offset else b"")            # field 4 offset (omitted when 0)
                + bytes([0x28]) + varint(size)                                    # field 5 size
                + bytes([0x35]) + struct.pack("<I", masked(crc32c(raw))))        # field 6 crc32c, fixed32: MASKED crc of the tensor bytes
    e_bias = entry([3], 0, 12, bias)
    e_kernel = entry([2, 3], 12, 24, kernel)

    # ================= the data block, uncompressed form [T] ==========================================================
    # entry := varint shared | varint non_shared | varint value_len | key[shared:] | value ; keys ascending
    k0, k1, k2 = b"", b"a/bias/.ATTRIBUTES/VARIABLE_VALUE", b"a/kernel/.ATTRIBUTES/VARIABLE_VALUE"
    assert k0 < k1 < k2
    ent0 = varint(0) + varint(len(k0)) + varint(len(header)) + k0 + header        # restart point 0: shared = 0
    ent1 = varint(0) + varint(len(k1)) + varint(len(e_bias)) + k1 + e_bias        # restart point 1: shared = 0 (full key)
    ent2 = varint(2) + varint(len(k2) - 2) + varint(len(e_kernel)) + k2[2:] + e_kernel   # shares "a/" with the previous key
    restarts = struct.pack("<II", 0, len(ent0)) + struct.pack("<I", 2)           # uint32 offsets of the restart points, then their count
    U = ent0 + ent1 + ent2 + restarts

    # ================= the same block as a snappy stream [S] ==========================================================
    tail = b"/.ATTRIBUTES/VARIABLE_VALUE"                                         # 27 bytes, occurs in ent1's key and again in ent2's
    p1 = U.index(tail)
    p2 = U.index(tail, p1 + 1)
    lit1, lit3 = U[:p2], U[p2 + len(tail):]
    assert 60 < len(lit1) <= 256 and len(lit3) <= 60                              # one literal of each length class
    comp = (varint(len(U))                                                        # preamble: uncompressed length
            + bytes([(60 << 2) | 0, len(lit1) - 1]) + lit1                        # literal, length 61..256: tag 60 << 2 | 00b, then (len - 1) in one byte
            + bytes([((len(tail) - 1) << 2) | 2]) + struct.pack("<H", p2 - p1)   # copy, 2-byte offset: tag = (len - 1) << 2 | 10b, offset LE
            + bytes([((len(lit3) - 1) << 2) | 0]) + lit3)                         # literal, length 1..60: tag = (len - 1) << 2 | 00b
    # executing the three elements forward reproduces U (this is the definition of the format, not a decoder under test)
    out = bytearray(lit1)
    for _ in range(len(tail)):
        out.append(out[-(p2 - p1)])
    out += lit3
    assert bytes(out) == U and len(comp) < len(U)

    def with_trailer(block, ctype):                                               # [T] block trailer: type byte + masked crc32c(block + type)
        return block + bytes([ctype]) + struct.pack("<I", masked(crc32c(block + bytes([ctype]))))

    # ================= file layout [T]: data block | metaindex block | index block | footer ===========================
    f = bytearray()
    data_handle = (len(f), len(comp))
    f += with_trailer(comp, 1)                                                    # compression type 1 = snappy
    meta = struct.pack("<II", 0, 1)                                               # empty block: restart array [0], count 1
    meta_handle = (len(f), len(meta))
    f += with_trailer(meta, 0)
    # index block: one entry per data block; key >= every key of that block (for the last block LevelDB stores a short successor of
    # its last key: first byte incremented -> "b"); value = BlockHandle = varint offset, varint size
    hv = varint(data_handle[0]) + varint(data_handle[1])
    ik = b"b"
    index = varint(0) + varint(len(ik)) + varint(len(hv)) + ik + hv + struct.pack("<II", 0, 1)
    index_handle = (len(f), len(index))
    f += with_trailer(index, 0)
    footer = varint(meta_handle[0]) + varint(meta_handle[1]) + varint(index_handle[0]) + varint(index_handle[1])
    footer += bytes(40 - len(footer))                                             # handles padded to 2 * kMaxEncodedLength = 40 bytes
    footer += struct.pack("<Q", 0xDB4775248B80FB57)                               # magic, little-endian
    assert len(footer) == 48
    f += footer

    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "variables.index"), "wb") as fh:
        fh.write(bytes(f))
    with open(os.path.join(OUT, "variables.data-00000-of-00001"), "wb") as fh:  # [B] DataFilename(prefix, shard 0, num_shards 1)
        fh.write(data_file)
    expected = {
        "index_sha1": hashlib.sha1(bytes(f)).hexdigest(), "data_sha1": hashlib.sha1(data_file).hexdigest(),
        "tensors": {
            k1.decode(): {"dtype": "float32", "shape": [3], "values": list(struct.unpack("<3f", bias)), "crc32c": crc32c(bias),
                          "sha1": hashlib.sha1(bias).hexdigest()},
            k2.decode(): {"dtype": "float32", "shape": [2, 3], "values": list(struct.unpack("<6f", kernel)), "crc32c": crc32c(kernel),
                          "sha1": hashlib.sha1(kernel).hexdigest()},
        },
        "block": {"uncompressed_bytes": len(U), "compressed_bytes": len(comp), "restarts": [0, len(ent0)], "copy_offset": p2 - p1},
    }
    with open(os.path.join(OUT, "expected.json"), "w") as fh:
        json.dump(expected, fh, indent=1)
    print(f"wrote {OUT}: index {len(f)} bytes (block {len(U)} -> {len(comp)} snappy), data {len(data_file)} bytes")


if __name__ == "__main__":
    main()
