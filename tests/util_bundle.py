"""Minimal, independent WRITER of TensorFlow tensor bundles (variables.index + variables.data-*) used to test
multilingual_kws_amd/checkpoint_import.py without TensorFlow.  Written from the format descriptions
(tensorflow/core/lib/io/format.cc, table_builder.cc, block_builder.cc; tensor_bundle.proto;
trackable_object_graph.proto), sharing no code with the reader except the CRC table."""
import struct

import numpy as np

from multilingual_kws_amd.checkpoint_import import crc32c, mask_crc

DT = {np.dtype("float32"): 1, np.dtype("float64"): 2, np.dtype("int32"): 3, np.dtype("int64"): 9}


def varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def field_varint(num, v):
    return varint(num << 3) + varint(v)


def field_bytes(num, b):
    return varint((num << 3) | 2) + varint(len(b)) + bytes(b)


def field_fixed32(num, v):
    return varint((num << 3) | 5) + struct.pack("<I", v)


def snappy_literals(raw):
    """A valid snappy stream that uses literal elements only."""
    out = bytearray(varint(len(raw)))
    pos = 0
    while pos < len(raw):
        chunk = raw[pos:pos + 60000]
        n = len(chunk) - 1
        if n < 60:
            out.append(n << 2)
        elif n < 256:
            out += bytes([60 << 2, n])
        else:
            out += bytes([61 << 2]) + struct.pack("<H", n)
        out += chunk
        pos += len(chunk)
    return bytes(out)


def build_block(entries, restart_interval):
    out, restarts, last = bytearray(), [], b""
    for i, (k, v) in enumerate(entries):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(k), len(last)) and k[shared] == last[shared]:
                shared += 1
        out += varint(shared) + varint(len(k) - shared) + varint(len(v)) + k[shared:] + v
        last = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def write_table(path, items, block_entries=7, restart_interval=4, compress=False):
    """items: {key bytes: value bytes}.  Several data blocks, prefix compression, optional snappy(-literal) blocks."""
    keys = sorted(items)
    f = bytearray()
    index = []

    def emit(block):
        payload, ctype = (snappy_literals(block), 1) if compress else (block, 0)
        off = len(f)
        f.extend(payload)
        f.append(ctype)
        f.extend(struct.pack("<I", mask_crc(crc32c(payload + bytes([ctype])))))
        return varint(off) + varint(len(payload))
    for s in range(0, len(keys), block_entries):
        chunk = keys[s:s + block_entries]
        index.append((chunk[-1], emit(build_block([(k, items[k]) for k in chunk], restart_interval))))
    meta = emit(build_block([], 1))
    idx = emit(build_block(index, 1))
    footer = meta + idx
    footer += b"\0" * (40 - len(footer)) + struct.pack("<Q", 0xDB4775248B80FB57)
    f.extend(footer)
    open(path, "wb").write(bytes(f))


def write_bundle(prefix, tensors, object_graph=None, num_shards=1, **table_kw):
    """tensors: {checkpoint key: ndarray}; object_graph: serialized TrackableObjectGraph bytes (optional); num_shards > 1 deals the
    tensors round-robin over data shards (what tf.train.Checkpoint writes from several devices)."""
    data = [bytearray() for _ in range(num_shards)]
    items = {b"": field_varint(1, num_shards) + field_varint(2, 0) + field_bytes(3, field_varint(1, 1))}

    def entry(dtype, shape, shard, off, size, crc):
        shp = b"".join(field_bytes(2, field_varint(1, d)) for d in shape)
        return field_varint(1, dtype) + field_bytes(2, shp) + field_varint(3, shard) + field_varint(4, off) + field_varint(5, size) + field_fixed32(6, crc)
    for i, k in enumerate(sorted(tensors)):
        a = np.ascontiguousarray(tensors[k])
        raw = a.astype(a.dtype.newbyteorder("<")).tobytes()
        sh = i % num_shards
        items[k.encode()] = entry(DT[a.dtype], a.shape, sh, len(data[sh]), len(raw), mask_crc(crc32c(raw)))
        data[sh] += raw
    if object_graph is not None:
        lens = varint(len(object_graph))
        raw = lens + struct.pack("<I", mask_crc(crc32c(struct.pack("<Q", len(object_graph))))) + object_graph
        items[b"_CHECKPOINTABLE_OBJECT_GRAPH"] = entry(7, (), 0, len(data[0]), len(raw), 0)
        data[0] += raw
    for sh in range(num_shards):
        open(prefix + f".data-{sh:05d}-of-{num_shards:05d}", "wb").write(bytes(data[sh]))
    write_table(prefix + ".index", items, **table_kw)


def keras_object_graph(layers, with_full_names=True, name_prefix=""):
    """TrackableObjectGraph of a flat Keras functional model.  layers: [(layer name, [variable leaf names])] in
    layer order.  Returns (serialized graph, {"<layer>/<leaf>": checkpoint key})."""
    nodes = [[[], []]]                     # node 0 = the model: [children, attributes]
    keys = {}
    for i, (lname, leaves) in enumerate(layers):
        lid = len(nodes)
        nodes.append([[], []])
        nodes[0][0].append((lid, f"layer_with_weights-{i}"))
        for leaf in leaves:
            vid = len(nodes)
            key = f"layer_with_weights-{i}/{leaf}/.ATTRIBUTES/VARIABLE_VALUE"
            full = f"{name_prefix}{lname}/{leaf}" if with_full_names else ""
            nodes.append([[], [("VARIABLE_VALUE", full, key)]])
            nodes[lid][0].append((vid, leaf))
            keys[f"{lname}/{leaf}"] = key
        kid = len(nodes)                   # every Keras layer also tracks non-variable children
        nodes.append([[], []])
        nodes[lid][0].append((kid, "keras_api"))
    out = b""
    for children, attrs in nodes:
        nb = b"".join(field_bytes(1, field_varint(1, c) + field_bytes(2, n.encode())) for c, n in children)
        nb += b"".join(field_bytes(2, field_bytes(1, a.encode()) + field_bytes(2, f.encode()) + field_bytes(3, k.encode())) for a, f, k in attrs)
        out += field_bytes(1, nb)
    return out, keys
