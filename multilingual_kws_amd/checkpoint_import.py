"""TensorFlow-free importer for the checkpoint every reference caller loads: the Keras SavedModel directory
`multilingual_context_73_0.8011` (docker/Dockerfile:69-70; transfer_learning.py:36 `tf.keras.models.load_model`).

A SavedModel keeps its variables in a TensorFlow *tensor bundle*, `variables/variables.index` +
`variables/variables.data-0000N-of-0000M` (tensorflow/core/util/tensor_bundle/tensor_bundle.{h,cc}):

  * `.index` is an SSTable in TensorFlow's port of the LevelDB table format (tensorflow/core/lib/io/table*.cc,
    format.cc): data blocks of prefix-compressed (shared, non_shared, value_len, key delta, value) entries with a
    restart array, each block followed by a 5-byte trailer (compression type: 0 none / 1 snappy; masked CRC32C), an
    index block mapping last-keys to block handles, and a 48-byte footer (metaindex handle, index handle, magic
    0xdb4775248b80fb57).
  * key "" holds a BundleHeaderProto (num_shards, endianness, version); every other key is a checkpoint key whose
    value is a BundleEntryProto (dtype, shape, shard_id, offset, size, crc32c) pointing into a data shard
    (tensorflow/core/protobuf/tensor_bundle.proto).
  * TF2 (object-based) checkpoints name variables by their path in the object graph, e.g.
    `layer_with_weights-3/kernel/.ATTRIBUTES/VARIABLE_VALUE`; key `_CHECKPOINTABLE_OBJECT_GRAPH` holds a serialized
    TrackableObjectGraph (tensorflow/core/protobuf/trackable_object_graph.proto) whose SerializedTensor entries
    carry both the checkpoint key and the variable's `full_name` ("stem_conv/kernel", "block2a_expand_bn/gamma", ...).

`load_savedmodel(dir)` returns {Keras variable name: ndarray}; `import_savedmodel(dir)` the weight blob of
multilingual_kws_amd.weights (names = mkws_embed_weight_manifest).  Variables are matched by `full_name` (nested
model prefixes and ":0" dropped); when an object graph has no usable full names, positionally: weighted layers in
object-graph order against the manifest's layer order (identical to Keras' layer order for this architecture).

HONEST STATUS: the released checkpoint is a GitHub release asset and cannot be fetched in this environment, and
TensorFlow cannot be installed, so this reader is verified against (a) tests/golden/tf_bundle/, a bundle assembled byte
by byte from the format specifications by a script that shares no code with this module or with the test writer
(tests/golden/make_tf_bundle_fixture.py), (b) bundles written by the minimal writer in tests/util_bundle.py (same
author as this reader: a self-comparison, kept for coverage of the object graph and multi-shard layouts), (c) snappy /
CRC-32C known-answer vectors -- NOT yet against `multilingual_context_73_0.8011` itself.
`python tools/import_savedmodel.py --verify <dir>` prints per-tensor shape / CRC verdict / sha1 for whoever holds it.
"""
import os
import struct

import numpy as np

TABLE_MAGIC = 0xDB4775248B80FB57
OBJECT_GRAPH_KEY = "_CHECKPOINTABLE_OBJECT_GRAPH"
ATTR_SUFFIX = "/.ATTRIBUTES/VARIABLE_VALUE"
# tensorflow/core/framework/types.proto
DTYPES = {1: np.dtype("<f4"), 2: np.dtype("<f8"), 3: np.dtype("<i4"), 4: np.dtype("u1"), 5: np.dtype("<i2"), 6: np.dtype("i1"),
          9: np.dtype("<i8"), 10: np.dtype("?"), 17: np.dtype("<u2"), 19: np.dtype("<f2"), 22: np.dtype("<u4"), 23: np.dtype("<u8")}
DT_STRING = 7


class CheckpointFormatError(ValueError):
    pass


# ---- CRC32C (Castagnoli), TF's mask ---------------------------------------------------------------------------------
def _crc_table():
    t = []
    for n in range(256):
        c = n
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        t.append(c)
    return t


_CRC = _crc_table()


_CRC_NP = np.asarray(_CRC, dtype=np.uint32)
_ZERO_OPS = {}


def _crc_scalar(data, c):
    for b in data:
        c = _CRC[(c ^ b) & 0xFF] ^ (c >> 8)
    return c


def _zero_op(L):
    """Columns of the GF(2)-linear map "advance the CRC register over L zero bytes" (one uint32 per input bit)."""
    if L not in _ZERO_OPS:
        st = (np.uint32(1) << np.arange(32, dtype=np.uint32)).astype(np.uint32)
        for _ in range(L):
            st = _CRC_NP[st & np.uint32(0xFF)] ^ (st >> np.uint32(8))
        _ZERO_OPS[L] = [int(v) for v in st]
    return _ZERO_OPS[L]


def crc32c(data, crc=0):
    """CRC-32C of `data`, continuing from `crc`.  Tensors are megabytes, so long inputs are cut into equal chunks whose
    register remainders are computed side by side in numpy (the register update is linear over GF(2):
    state(s0, A || B) = Z_len(B)(state(s0, A)) xor state(0, B)) and folded in order; short inputs take the byte loop."""
    data = bytes(data)
    c = crc ^ 0xFFFFFFFF
    L = 1024
    n = len(data) // L
    if n >= 8:
        a = np.frombuffer(data, dtype=np.uint8, count=n * L).reshape(n, L)
        st = np.zeros(n, dtype=np.uint32)
        for j in range(L):
            st = _CRC_NP[(st ^ a[:, j]) & np.uint32(0xFF)] ^ (st >> np.uint32(8))
        cols = _zero_op(L)
        for r in st.tolist():
            z = 0
            k = 0
            while c:
                if c & 1:
                    z ^= cols[k]
                c >>= 1
                k += 1
            c = z ^ r
        data = data[n * L:]
    return _crc_scalar(data, c) ^ 0xFFFFFFFF


def mask_crc(c):
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


# ---- varints / protobuf wire format ---------------------------------------------------------------------------------
def read_varint(buf, pos):
    out = shift = 0
    while True:
        if pos >= len(buf):
            raise CheckpointFormatError("truncated varint")
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7
        if shift > 63:
            raise CheckpointFormatError("varint too long")


def parse_proto(buf):
    """Wire-level parse: [(field number, wire type, value)]; value = int (varint / fixed) or bytes (length-delimited)."""
    out, pos = [], 0
    buf = bytes(buf)
    while pos < len(buf):
        tag, pos = read_varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = read_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            n, pos = read_varint(buf, pos)
            if pos + n > len(buf):
                raise CheckpointFormatError("truncated length-delimited field")
            v = buf[pos:pos + n]
            pos += n
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise CheckpointFormatError(f"unsupported protobuf wire type {wt}")
        out.append((field, wt, v))
    return out


def _signed64(v):
    return v - (1 << 64) if v >= (1 << 63) else v


# ---- snappy (raw format: varint length, then literal / copy elements) ------------------------------------------------
def snappy_decompress(buf):
    buf = bytes(buf)
    n, pos = read_varint(buf, 0)
    out = bytearray()
    while pos < len(buf):
        tag = buf[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:                                   # literal
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(buf[pos:pos + nb], "little")
                pos += nb
            ln += 1
            out += buf[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:                                   # copy, 1-byte offset
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | buf[pos]
            pos += 1
        elif kind == 2:                                 # copy, 2-byte offset
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 2], "little")
            pos += 2
        else:                                           # copy, 4-byte offset
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 4], "little")
            pos += 4
        if off == 0 or off > len(out):
            raise CheckpointFormatError("snappy copy reaches before the start of the output")
        for _ in range(ln):                             # byte-wise: copies may overlap their own output
            out.append(out[-off])
    if len(out) != n:
        raise CheckpointFormatError(f"snappy stream decodes to {len(out)} bytes, header says {n}")
    return bytes(out)


# ---- SSTable ---------------------------------------------------------------------------------------------------------
def _read_block(data, offset, size, verify=True):
    raw = data[offset:offset + size]
    if len(raw) != size or offset + size + 5 > len(data):
        raise CheckpointFormatError("block handle points outside the file")
    ctype = data[offset + size]
    if verify:
        stored = struct.unpack_from("<I", data, offset + size + 1)[0]
        if mask_crc(crc32c(data[offset:offset + size + 1])) != stored:
            raise CheckpointFormatError("block checksum mismatch")
    if ctype == 0:
        return raw
    if ctype == 1:
        return snappy_decompress(raw)
    raise CheckpointFormatError(f"unknown block compression type {ctype}")


def _block_entries(block):
    if len(block) < 4:
        raise CheckpointFormatError("block too small")
    nrestarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * nrestarts
    if end < 0:
        raise CheckpointFormatError("bad restart array")
    pos, key = 0, b""
    while pos < end:
        shared, pos = read_varint(block, pos)
        non_shared, pos = read_varint(block, pos)
        vlen, pos = read_varint(block, pos)
        if shared > len(key) or pos + non_shared + vlen > end:
            raise CheckpointFormatError("corrupt block entry")
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        yield key, block[pos:pos + vlen]
        pos += vlen


def read_table(path, verify=True):
    """{key bytes: value bytes} of a TensorFlow / LevelDB-format SSTable file."""
    data = open(path, "rb").read()
    if len(data) < 48 or struct.unpack_from("<Q", data, len(data) - 8)[0] != TABLE_MAGIC:
        raise CheckpointFormatError(f"{path}: not a TensorFlow table file (bad magic)")
    footer = data[-48:]
    _, pos = read_varint(footer, 0)          # metaindex handle (unused)
    _, pos = read_varint(footer, pos)
    ioff, pos = read_varint(footer, pos)
    isize, pos = read_varint(footer, pos)
    out = {}
    for _, handle in _block_entries(_read_block(data, ioff, isize, verify)):
        boff, p = read_varint(handle, 0)
        bsize, _ = read_varint(handle, p)
        for k, v in _block_entries(_read_block(data, boff, bsize, verify)):
            out[bytes(k)] = bytes(v)
    return out


# ---- tensor bundle ---------------------------------------------------------------------------------------------------
class BundleReader:
    """Reads `<prefix>.index` + `<prefix>.data-XXXXX-of-YYYYY`."""

    def __init__(self, prefix, verify=True):
        self.prefix = prefix
        table = read_table(prefix + ".index", verify)
        if b"" not in table:
            raise CheckpointFormatError("bundle has no header entry")
        hdr = {f: v for f, _, v in parse_proto(table[b""])}
        self.num_shards = hdr.get(1, 1)
        if hdr.get(2, 0) != 0:
            raise CheckpointFormatError("big-endian bundles are not supported")
        self.entries = {}
        for k, v in table.items():
            if k == b"":
                continue
            e = {"dtype": 0, "shape": [], "shard_id": 0, "offset": 0, "size": 0, "crc32c": None, "sliced": False}
            for f, _, val in parse_proto(v):
                if f == 1:
                    e["dtype"] = val
                elif f == 2:
                    for f2, _, dim in parse_proto(val):
                        if f2 == 2:
                            e["shape"].append(_signed64(dict((a, c) for a, _, c in parse_proto(dim)).get(1, 0)))
                elif f == 3:
                    e["shard_id"] = val
                elif f == 4:
                    e["offset"] = val
                elif f == 5:
                    e["size"] = val
                elif f == 6:
                    e["crc32c"] = val
                elif f == 7:
                    e["sliced"] = True
            self.entries[k.decode("utf-8")] = e
        self._shards = {}

    def keys(self):
        return sorted(self.entries)

    def _shard(self, i):
        if i not in self._shards:
            path = f"{self.prefix}.data-{i:05d}-of-{self.num_shards:05d}"
            self._shards[i] = np.memmap(path, dtype=np.uint8, mode="r")
        return self._shards[i]

    def raw(self, key):
        e = self.entries[key]
        if e["sliced"]:
            raise CheckpointFormatError(f"{key}: partitioned (sliced) variables are not supported")
        buf = self._shard(e["shard_id"])[e["offset"]:e["offset"] + e["size"]]
        if buf.shape[0] != e["size"]:
            raise CheckpointFormatError(f"{key}: data shard is truncated")
        return buf

    def tensor(self, key, verify_crc=False):
        e = self.entries[key]
        buf = self.raw(key)
        if e["dtype"] == DT_STRING:
            b = bytes(buf)
            n = int(np.prod(e["shape"])) if e["shape"] else 1
            lens, pos = [], 0
            for _ in range(n):
                ln, pos = read_varint(b, pos)
                lens.append(ln)
            pos += 4                                       # masked crc32c of the lengths
            out = []
            for ln in lens:
                out.append(b[pos:pos + ln])
                pos += ln
            return out[0] if not e["shape"] else out
        if e["dtype"] not in DTYPES:
            raise CheckpointFormatError(f"{key}: unsupported dtype enum {e['dtype']}")
        if verify_crc and e["crc32c"] is not None and mask_crc(crc32c(bytes(buf))) != e["crc32c"]:
            raise CheckpointFormatError(f"{key}: tensor checksum mismatch")
        dt = DTYPES[e["dtype"]]
        arr = np.frombuffer(bytes(buf), dtype=dt)
        if arr.size != int(np.prod(e["shape"])) if e["shape"] else arr.size != 1:
            raise CheckpointFormatError(f"{key}: {arr.size} elements do not fill shape {e['shape']}")
        return arr.reshape(e["shape"])

    def object_graph(self):
        """[(node children [(child id, local name)], attributes [(name, full_name, checkpoint_key)])] or None."""
        if OBJECT_GRAPH_KEY not in self.entries:
            return None
        nodes = []
        for f, _, nb in parse_proto(self.tensor(OBJECT_GRAPH_KEY)):
            if f != 1:
                continue
            children, attrs = [], []
            for f2, _, v in parse_proto(nb):
                if f2 == 1:
                    d = dict((a, c) for a, _, c in parse_proto(v))
                    children.append((d.get(1, 0), d.get(2, b"").decode("utf-8")))
                elif f2 == 2:
                    d = dict((a, c) for a, _, c in parse_proto(v))
                    attrs.append((d.get(1, b"").decode("utf-8"), d.get(2, b"").decode("utf-8"), d.get(3, b"").decode("utf-8")))
            nodes.append((children, attrs))
        return nodes


# ---- Keras naming ----------------------------------------------------------------------------------------------------
def _short(full_name):
    parts = full_name.split(":")[0].split("/")
    return "/".join(parts[-2:])


def _layer_order_variables(nodes):
    """Variables in Keras layer order: DFS over `layer_with_weights-K` children (K ascending), each layer's own
    variables (object-graph children that own a VARIABLE_VALUE attribute) in the order the layer lists them."""
    order, seen = [], set()

    def visit(i):
        if i in seen:
            return
        seen.add(i)
        children, _ = nodes[i]
        layers = sorted(((int(n.rsplit("-", 1)[1]), c) for c, n in children if n.startswith("layer_with_weights-")), key=lambda t: t[0])
        if layers:
            for _, c in layers:
                visit(c)
            return
        for c, n in children:                                  # a leaf layer: its variables
            for aname, full, key in nodes[c][1]:
                if aname == "VARIABLE_VALUE":
                    order.append((n, full, key))
    visit(0)
    return order


def load_savedmodel(path, verify=True):
    """{Keras variable name ("stem_conv/kernel", ...): float32 ndarray} from a SavedModel directory, a
    `variables/` directory or a bundle prefix.  Optimizer slots and the classifier layer beyond dense_2 are returned
    too when present (callers pick what they need)."""
    prefix = path
    for cand in (os.path.join(path, "variables", "variables"), os.path.join(path, "variables"), path):
        if os.path.exists(cand + ".index"):
            prefix = cand
            break
    else:
        raise FileNotFoundError(f"no variables.index under {path}")
    rd = BundleReader(prefix, verify)
    nodes = rd.object_graph()
    named, positional = {}, []
    if nodes is not None:
        for _, attrs in nodes:
            for aname, full, key in attrs:
                if aname == "VARIABLE_VALUE" and full and key in rd.entries and "/.OPTIMIZER_SLOT/" not in key:
                    named.setdefault(_short(full), key)
        positional = [(n, key) for n, _, key in _layer_order_variables(nodes) if key in rd.entries]
    else:                                                       # TF1-style name-based checkpoint: keys are variable names
        for k in rd.keys():
            named.setdefault(_short(k), k)
    return {"reader": rd, "named": named, "positional": positional}


def import_savedmodel(path, verify=True):
    """SavedModel directory -> weight blob of multilingual_kws_amd.weights (validated against the manifest)."""
    from . import weights
    info = load_savedmodel(path, verify)
    rd, named = info["reader"], info["named"]
    tensors = weights.manifest()
    have_all = all(t["name"] in named for t in tensors)
    out = {}
    if have_all:
        for t in tensors:
            out[t["name"]] = rd.tensor(named[t["name"]], verify_crc=verify)
    else:
        # positional fallback: group the manifest by layer, walk the checkpoint's weighted layers in order
        layers, cur = [], None
        for t in tensors:
            lname = t["name"].split("/")[0]
            if lname != cur:
                layers.append([])
                cur = lname
            layers[-1].append(t)
        seq = info["positional"]
        groups, last = [], None
        for local, key in seq:
            base = key[:-len(ATTR_SUFFIX)] if key.endswith(ATTR_SUFFIX) else key
            owner = base.rsplit("/", 1)[0]
            if owner != last:
                groups.append({})
                last = owner
            groups[-1][local] = key
        if len(groups) < len(layers):
            raise CheckpointFormatError(f"checkpoint has {len(groups)} weighted layers, the embedding needs {len(layers)} "
                                        f"(and no usable variable names were found: missing e.g. {[t['name'] for t in tensors if t['name'] not in named][:3]})")
        for lay, grp in zip(layers, groups):
            for t in lay:
                leaf = t["name"].split("/")[1]
                if leaf not in grp:
                    raise CheckpointFormatError(f"layer {t['name'].split('/')[0]}: checkpoint layer has {sorted(grp)}, no '{leaf}'")
                out[t["name"]] = rd.tensor(grp[leaf], verify_crc=verify)
    for t in tensors:
        v = np.asarray(out[t["name"]], dtype=np.float32)
        if t["name"].startswith("normalization/") and v.size == 1:
            v = v.reshape(t["shape"])
        out[t["name"]] = v
    return weights.from_named_tensors(out)


# =====================================================================================================================
# Keras `.h5` checkpoints (tf.keras.models.load_model / load_weights accept them: transfer_learning.py:36) -- HDF5 without libhdf5.
#
# What h5py / HDF5 1.8-1.10 write with default settings (libver "earliest"), and all this reader accepts (HDF5 File Format
# Specification 2.0, sections III and IV):
#   superblock version 0 / 1 (8-byte offsets and lengths) -> root symbol-table entry -> object headers VERSION 1 (16-byte prefix,
#   8-byte aligned messages, continuation blocks) -> old-style groups (symbol table message -> v1 B-tree of SNOD nodes, names in a
#   local heap) -> datasets with a CONTIGUOUS or COMPACT layout, fixed-point / floating-point / fixed-length string datatypes ->
#   attributes (message versions 1-3) holding fixed-length strings, variable-length strings in the global heap (what h5py 3 writes for
#   Keras' layer_names / weight_names lists) or numbers.
# Refused loudly (CheckpointFormatError), never guessed: superblock 2 / 3 and version-2 object headers ("OHDR", libver "latest"),
# chunked / compressed / virtual layouts, variable-length or compound datatypes of DATASETS, external files, shared messages.
# Keras' layout on top of it (keras/saving/hdf5_format.py): see tests/golden/make_h5_fixture.py.
# Verified against files written by h5py (HDF5 1.10.6), tests/golden/keras_h5/ -- a writer this project did not write.
# =====================================================================================================================
HDF5_SIGNATURE = b"\x89HDF\r\n\x1a\n"
_UNDEF = 0xFFFFFFFFFFFFFFFF


def _format_errors(fn):
    """A corrupt or truncated file must fail as CheckpointFormatError, not as whatever the first out-of-range unpack / slice / search raises."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *a, **k):
        try:
            return fn(self, *a, **k)
        except CheckpointFormatError:
            raise
        except (struct.error, IndexError, ValueError, UnicodeDecodeError, OverflowError, RecursionError) as exc:
            raise CheckpointFormatError(f"corrupt or truncated HDF5 file ({fn.__name__}: {exc!r})") from exc
    return wrapped


class H5File:
    """Minimal read-only view of an HDF5 file: groups as {name: object-header address}, datasets as numpy arrays, attributes."""

    def __init__(self, path):
        with open(path, "rb") as f:
            self.buf = f.read()
        b = self.buf
        base = None
        for off in [0] + [512 << k for k in range(0, 12)]:
            if b[off:off + 8] == HDF5_SIGNATURE:
                base = off
                break
        if base is None:
            raise CheckpointFormatError(f"{path}: no HDF5 signature")
        ver = b[base + 8]
        if ver not in (0, 1):
            raise CheckpointFormatError(f"{path}: HDF5 superblock version {ver} (written with libver='latest'?): only versions 0 and 1 are read; "
                                        "re-save with h5py's default libver or convert with `h5repack --low=EARLIEST --high=V18`")
        so, sl = b[base + 13], b[base + 14]
        if (so, sl) != (8, 8):
            raise CheckpointFormatError(f"{path}: {so}-byte offsets / {sl}-byte lengths (only 8 / 8 are read)")
        p = base + 24 + (4 if ver == 1 else 0)
        self.base_addr, _free, self.eof, _drv = struct.unpack_from("<4Q", b, p)
        p += 32
        _name_off, root_oh, cache, _r = struct.unpack_from("<QQII", b, p)
        self.root = root_oh
        # OBJECT addresses in the file are relative to the base address (a user block in front of the superblock moves it: _span);
        # the END-OF-FILE address is stored absolute (libhdf5 writes rel_eoa + base_addr: a file h5py wrote with userblock_size=512
        # has base 512 and eof == its size), so it is compared with the file length as it stands
        if self.eof > len(b):
            raise CheckpointFormatError(f"{path}: truncated (end-of-file address {self.eof}, file has {len(b)} bytes)")

    def _span(self, addr, size, what):
        """Absolute offset of the file address `addr`, checked to hold `size` bytes."""
        a = addr + self.base_addr
        if addr < 0 or size < 0 or a + size > len(self.buf):
            raise CheckpointFormatError(f"{what}: address {addr} + {size} bytes lies outside the file ({len(self.buf)} bytes)")
        return a

    # ---- object headers ------------------------------------------------------------------------------------------
    @_format_errors
    def messages(self, addr):
        """[(type, flags, payload bytes)] of the version-1 object header at addr (continuation blocks followed)."""
        b = self.buf
        a = self._span(addr, 16, "object header")
        if b[a:a + 4] == b"OHDR":
            raise CheckpointFormatError("version-2 object header (file written with libver='latest'): not read")
        if b[a] != 1:
            raise CheckpointFormatError(f"object header version {b[a]} at {addr}")
        nmsg, _ref, hsize = struct.unpack_from("<HII", b, a + 2)
        blocks = [(a + 16, hsize)]
        out = []
        while blocks and len(out) < nmsg:
            p, left = blocks.pop(0)
            end = p + left
            while p + 8 <= end and len(out) < nmsg:
                mtype, msize, mflags = struct.unpack_from("<HHB", b, p)
                payload = b[p + 8:p + 8 + msize]
                if mflags & 0x02:
                    raise CheckpointFormatError(f"shared object-header message (type {mtype:#x}): not read")
                if mtype == 0x0010:                                  # continuation: (address, length)
                    ca, cl = struct.unpack_from("<QQ", payload, 0)
                    blocks.append((self._span(ca, cl, "object-header continuation block"), cl))
                out.append((mtype, mflags, payload))
                p += 8 + msize
        return out

    # ---- groups --------------------------------------------------------------------------------------------------
    def _heap_string(self, heap_addr, off):
        b = self.buf
        h = heap_addr + self.base_addr
        if b[h:h + 4] != b"HEAP":
            raise CheckpointFormatError("bad local heap signature")
        data_addr = struct.unpack_from("<Q", b, h + 24)[0] + self.base_addr
        e = b.index(b"\x00", data_addr + off)
        return b[data_addr + off:e].decode("utf-8")

    @_format_errors
    def _btree_group(self, node_addr, heap_addr, out):
        b = self.buf
        a = self._span(node_addr, 8, "group B-tree / symbol-table node")
        if b[a:a + 4] == b"SNOD":
            nsym = struct.unpack_from("<H", b, a + 6)[0]
            for i in range(nsym):
                name_off, oh = struct.unpack_from("<QQ", b, a + 8 + 40 * i)
                out[self._heap_string(heap_addr, name_off)] = oh
            return
        if b[a:a + 4] != b"TREE":
            raise CheckpointFormatError(f"bad group B-tree node signature at {node_addr}")
        ntype, _level, used = struct.unpack_from("<BBH", b, a + 4)
        if ntype != 0:
            raise CheckpointFormatError("group B-tree node of type %d" % ntype)
        p = a + 24                                                   # after left / right sibling addresses
        for i in range(used):                                        # key_i (8), child_i (8), ..., key_used
            child = struct.unpack_from("<Q", b, p + 8 + 16 * i)[0]
            self._btree_group(child, heap_addr, out)

    @_format_errors
    def members(self, addr):
        """{link name: object-header address} of the group at addr, in name order (old-style groups only)."""
        for mtype, _f, pl in self.messages(addr):
            if mtype == 0x0011:
                btree, heap = struct.unpack_from("<QQ", pl, 0)
                out = {}
                self._btree_group(btree, heap, out)
                return out
            if mtype in (0x0002, 0x0006):
                raise CheckpointFormatError("new-style group (link messages): not read")
        return None                                                  # not a group

    # ---- datatypes / dataspaces ------------------------------------------------------------------------------------
    @staticmethod
    def _dtype(pl):
        cv, b0, _b1, _b2, size = struct.unpack_from("<BBBBI", pl, 0)
        cls, ver = cv & 0x0F, cv >> 4
        if ver not in (1, 2, 3):
            raise CheckpointFormatError(f"datatype message version {ver}")
        order = ">" if (b0 & 1) else "<"
        if cls == 0:                                                 # fixed point
            signed = bool(b0 & 0x08)
            return np.dtype(f"{order}{'i' if signed else 'u'}{size}")
        if cls == 1:                                                 # IEEE floating point (sizes 2, 4, 8)
            if size not in (2, 4, 8):
                raise CheckpointFormatError(f"{size}-byte floating-point type")
            return np.dtype(f"{order}f{size}")
        if cls == 3:                                                 # fixed-length string
            return np.dtype(f"S{size}")
        if cls == 9 and (b0 & 0x0F) == 1:                            # variable-length STRING: (length, global-heap address, index) per element
            return "vlen-string"
        names = {2: "time", 4: "bit field", 5: "opaque", 6: "compound", 7: "reference", 8: "enum", 9: "variable-length", 10: "array"}
        raise CheckpointFormatError(f"HDF5 datatype class {cls} ({names.get(cls, '?')}): not read")

    @staticmethod
    def _shape(pl):
        ver, rank, flags = pl[0], pl[1], pl[2]
        if ver == 1:
            p = 8
        elif ver == 2:
            if pl[3] == 2:
                return None                                          # null dataspace
            p = 4
        else:
            raise CheckpointFormatError(f"dataspace message version {ver}")
        return tuple(struct.unpack_from("<%dQ" % rank, pl, p)) if rank else ()

    def _global_heap_object(self, coll_addr, index):
        """Object `index` of the global heap collection at coll_addr (variable-length data lives there)."""
        b = self.buf
        a = coll_addr + self.base_addr
        if b[a:a + 4] != b"GCOL":
            raise CheckpointFormatError("bad global heap signature")
        size = struct.unpack_from("<Q", b, a + 8)[0]
        p, end = a + 16, a + size
        while p + 16 <= end:
            idx, _ref, _r, osz = struct.unpack_from("<HHIQ", b, p)
            if idx == 0:
                break                                                # free space: end of the objects
            if idx == index:
                return b[p + 16:p + 16 + osz]
            p += 16 + ((osz + 7) & ~7)
        raise CheckpointFormatError(f"global heap object {index} not found in the collection at {coll_addr}")

    # ---- datasets --------------------------------------------------------------------------------------------------
    @_format_errors
    def dataset(self, addr):
        dt = shape = layout = None
        for mtype, _f, pl in self.messages(addr):
            if mtype == 0x0001:
                shape = self._shape(pl)
            elif mtype == 0x0003:
                dt = self._dtype(pl)
            elif mtype == 0x0008:
                layout = pl
            elif mtype == 0x000B:
                raise CheckpointFormatError("dataset with a filter pipeline (compressed): not read")
            elif mtype == 0x0007:
                raise CheckpointFormatError("dataset stored in external files: not read")
        if dt is None or shape is None or layout is None:
            raise CheckpointFormatError(f"object at {addr} is not a dataset")
        if isinstance(dt, str):
            raise CheckpointFormatError("dataset of variable-length strings: not read")
        n = int(np.prod(shape, dtype=np.int64)) if shape else 1
        ver, cls = layout[0], layout[1]
        if ver != 3:
            raise CheckpointFormatError(f"data layout message version {ver} (only version 3 is read)")
        if cls == 0:                                                 # compact: data inside the header
            size = struct.unpack_from("<H", layout, 2)[0]
            raw = layout[4:4 + size]
        elif cls == 1:                                               # contiguous
            daddr, size = struct.unpack_from("<QQ", layout, 2)
            if daddr == _UNDEF:
                return np.zeros(shape, dt.newbyteorder("="))          # never written: fill value 0
            a = self._span(daddr, size, f"contiguous data of the dataset at {addr}")
            raw = self.buf[a:a + size]
        else:
            raise CheckpointFormatError("chunked dataset layout (chunking / compression): not read -- Keras writes contiguous datasets; "
                                        "h5repack -l CONTI converts")
        if len(raw) < n * dt.itemsize:
            raise CheckpointFormatError(f"dataset at {addr}: {len(raw)} bytes for {n} x {dt}")
        return np.frombuffer(raw, dtype=dt, count=n).reshape(shape).astype(dt.newbyteorder("="))

    # ---- attributes ------------------------------------------------------------------------------------------------
    @_format_errors
    def attributes(self, addr):
        """{name: ndarray} for the attributes this reader understands (others are skipped: Keras' JSON configs may be variable-length)."""
        out = {}
        for mtype, _f, pl in self.messages(addr):
            if mtype != 0x000C:
                continue
            ver = pl[0]
            nsz, dsz, ssz = struct.unpack_from("<HHH", pl, 2)
            if ver == 1:
                p = 8
                pad = lambda n: (n + 7) & ~7
            elif ver in (2, 3):
                p = 8 if ver == 2 else 9
                pad = lambda n: n
                if pl[1] & 0x03:
                    continue                                         # shared datatype / dataspace
            else:
                continue
            name = pl[p:p + nsz].split(b"\x00")[0].decode("utf-8")
            p += pad(nsz)
            try:
                dt = self._dtype(pl[p:p + dsz])
            except CheckpointFormatError:
                continue                                             # e.g. variable-length strings
            p += pad(dsz)
            shape = self._shape(pl[p:p + ssz])
            p += pad(ssz)
            if shape is None:
                continue
            n = int(np.prod(shape, dtype=np.int64)) if shape else 1
            if dt == "vlen-string":                                  # what h5py 3 writes for a list of (byte) strings
                vals = []
                for i in range(n):
                    ln, coll, idx = struct.unpack_from("<IQI", pl, p + 16 * i)
                    vals.append(self._global_heap_object(coll, idx)[:ln] if ln else b"")
                arr = np.empty(n, dtype=object)
                arr[:] = vals
                out[name] = arr.reshape(shape)
            else:
                out[name] = np.frombuffer(pl[p:p + n * dt.itemsize], dtype=dt, count=n).reshape(shape)
        return out

    def walk(self, addr, prefix=""):
        """Yields (path, object-header address) of every dataset below the group at addr."""
        mem = self.members(addr)
        if mem is None:
            yield prefix, addr
            return
        for name in mem:
            yield from self.walk(mem[name], f"{prefix}/{name}" if prefix else name)


def _attr_strings(attrs, name):
    """Keras' (possibly chunked: name0, name1, ...) list-of-strings attribute -> [str]."""
    parts = [attrs[name]] if name in attrs else []
    k = 0
    while not parts or k > 0:
        key = f"{name}{k}"
        if key not in attrs:
            break
        parts.append(attrs[key])
        k += 1
    out = []
    for a in parts:
        for s in np.atleast_1d(a).tolist():
            out.append(s.decode("utf-8") if isinstance(s, bytes) else str(s))
    return out


def load_h5(path):
    """Keras `.h5` (model.save or model.save_weights) -> {"named": {Keras variable name: ndarray}, "layers": [(layer, [weight names])]}.
    Variable names are the dataset names without ':0' ("stem_conv/kernel"); names inside a nested model keep their own prefix."""
    f = H5File(path)
    root = f.members(f.root)
    if root is None:
        raise CheckpointFormatError(f"{path}: root object is not a group")
    top = root["model_weights"] if "model_weights" in root else f.root
    attrs = f.attributes(top)
    members = f.members(top)
    layer_names = _attr_strings(attrs, "layer_names") or sorted(k for k, v in members.items() if f.members(v) is not None)
    named, layers = {}, []
    for lname in layer_names:
        if lname not in members:
            raise CheckpointFormatError(f"{path}: layer_names lists '{lname}' but the file has no such group")
        g = members[lname]
        wnames = _attr_strings(f.attributes(g), "weight_names")
        datasets = dict(f.walk(g))
        if not wnames:
            wnames = sorted(datasets)
        for w in wnames:
            if w not in datasets:
                raise CheckpointFormatError(f"{path}: layer '{lname}' lists weight '{w}' but holds {sorted(datasets)[:4]}...")
            named[_short(w)] = f.dataset(datasets[w])
        layers.append((lname, wnames))
    return {"named": named, "layers": layers}


def import_h5(path):
    """Keras `.h5` checkpoint -> weight blob of multilingual_kws_amd.weights (every manifest tensor must be present, shapes checked)."""
    from . import weights
    named = load_h5(path)["named"]
    out = {}
    for t in weights.manifest():
        if t["name"] not in named:
            raise CheckpointFormatError(f"{path}: no variable '{t['name']}' (file has {len(named)} variables, e.g. {sorted(named)[:3]})")
        v = np.asarray(named[t["name"]], dtype=np.float32)
        if t["name"].startswith("normalization/") and v.size == 1:
            v = v.reshape(t["shape"])
        if list(v.shape) != list(t["shape"]):
            raise CheckpointFormatError(f"{path}: '{t['name']}' has shape {list(v.shape)}, the embedding needs {t['shape']}")
        out[t["name"]] = v
    return weights.from_named_tensors(out)
