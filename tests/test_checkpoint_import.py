"""TensorFlow-free SavedModel variable-bundle importer (multilingual_kws_amd/checkpoint_import.py) against bundles
written by the independent minimal writer of tests/util_bundle.py, plus known-answer vectors for CRC32C and snappy.
NOT verified against the released multilingual_context_73_0.8011 (not fetchable here) -- see the module docstring."""
import os

import numpy as np
import pytest

from multilingual_kws_amd import checkpoint_import as ci, weights
from tests import util_bundle as ub


def test_crc32c_and_mask_known_answers():
    assert ci.crc32c(b"123456789") == 0xE3069283                       # the CRC-32C check value
    assert ci.crc32c(bytes(32)) == 0x8A9136AA                          # RFC 3720 B.4: 32 zero bytes
    assert ci.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43                 # RFC 3720 B.4: 32 bytes of 0xff
    assert ci.crc32c(b"6789", ci.crc32c(b"12345")) == 0xE3069283       # incremental use
    c = ci.crc32c(b"foo")
    assert ci.mask_crc(c) != c and ci.mask_crc(c) == ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def test_snappy_known_answer_vectors():
    # literal "abcd", then an overlapping 1-byte-offset copy (offset 4, length 8), then a 2-byte-offset copy (offset 12, length 5)
    stream = bytes([17]) + bytes([3 << 2]) + b"abcd" + bytes([((8 - 4) << 2) | 1 | (0 << 5), 4]) + bytes([((5 - 1) << 2) | 2, 12, 0])
    assert ci.snappy_decompress(stream) == b"abcdabcdabcd" + b"abcda"
    # run-length style: literal "x" + copy offset 1 length 10 (overlaps its own output)
    assert ci.snappy_decompress(bytes([11, 0, ord("x"), ((10 - 4) << 2) | 1, 1])) == b"x" * 11
    # long literal with a 2-byte length, as the writer emits
    raw = bytes(range(256)) * 5
    assert ci.snappy_decompress(ub.snappy_literals(raw)) == raw
    with pytest.raises(ci.CheckpointFormatError):
        ci.snappy_decompress(bytes([4, 0, ord("x"), (0 << 2) | 1, 9]))      # copy from before the start
    with pytest.raises(ci.CheckpointFormatError):
        ci.snappy_decompress(bytes([9, 0, ord("x")]))                        # shorter than its header says


def test_snappy_decoder_against_the_reference_library(golden_dir):
    """Streams compressed by Google's libsnappy 1.1.8 (tests/golden/make_snappy_fixture.py, ctypes; fixture = bytes in / bytes out):
    literals of every length class, 1- and 2-byte-offset copies, overlapping copies, multi-block inputs, SSTable-like blocks."""
    G = np.load(os.path.join(golden_dir, "snappy_vectors.npz"))
    names = [k[2:] for k in G.files if k.startswith("z/")]
    assert len(names) == 14
    for n in names:
        assert bytes(ci.snappy_decompress(G["z/" + n].tobytes())) == G["in/" + n].tobytes(), n
    z = G["z/text"].tobytes()
    with pytest.raises(ci.CheckpointFormatError):
        ci.snappy_decompress(z[:len(z) // 2])                    # truncated stream


@pytest.mark.parametrize("compress", [False, True])
@pytest.mark.parametrize("block_entries,restart", [(1, 1), (7, 4), (1000, 16)])
def test_table_round_trip(tmp_path, compress, block_entries, restart):
    rng = np.random.default_rng(0)
    items = {b"": b"hdr"}
    for i in range(83):
        items[f"layer_with_weights-{i // 3}/var{i % 3}/.ATTRIBUTES/VARIABLE_VALUE".encode()] = rng.bytes(int(rng.integers(0, 40)))
    path = str(tmp_path / "t.index")
    ub.write_table(path, items, block_entries=block_entries, restart_interval=restart, compress=compress)
    assert ci.read_table(path) == items
    raw = bytearray(open(path, "rb").read())
    raw[10] ^= 0x40                                                  # corrupt a data block
    open(path, "wb").write(bytes(raw))
    with pytest.raises(ci.CheckpointFormatError):
        ci.read_table(path)
    open(path, "wb").write(bytes(raw[:-3]))
    with pytest.raises(ci.CheckpointFormatError):
        ci.read_table(path)


def _keras_layers():
    """[(layer name, [leaf names])] of the embedding model in Keras layer order, from the manifest, + what else the released
    checkpoint holds: Normalization's count and the 761-way classifier the reference cuts off (transfer_learning.py:36-43)."""
    layers = []
    for t in weights.manifest():
        lname, leaf = t["name"].split("/")
        if not layers or layers[-1][0] != lname:
            layers.append((lname, []))
        layers[-1][1].append(leaf)
    layers[0][1].append("count")
    layers.append(("dense_3", ["kernel", "bias"]))
    return layers


def _write_model(tmp_path, blob, with_full_names, name_prefix="", **table_kw):
    layers = _keras_layers()
    graph, keys = ub.keras_object_graph(layers, with_full_names, name_prefix)
    tensors = {}
    for t in weights.manifest():
        tensors[keys[t["name"]]] = blob[t["offset"]:t["offset"] + t["count"]].reshape(t["shape"])
    tensors[keys["normalization/count"]] = np.asarray(0, dtype=np.int64)
    tensors[keys["dense_3/kernel"]] = np.zeros((1024, 761), np.float32)
    tensors[keys["dense_3/bias"]] = np.zeros(761, np.float32)
    tensors["optimizer/iter/.ATTRIBUTES/VARIABLE_VALUE"] = np.asarray(12345, dtype=np.int64)
    d = tmp_path / "saved_model" / "variables"
    os.makedirs(d)
    ub.write_bundle(str(d / "variables"), tensors, graph, **table_kw)
    return str(tmp_path / "saved_model")


def test_import_savedmodel_by_variable_name(tmp_path):
    blob = weights.synthetic_blob(seed=3, calibrate=False)
    path = _write_model(tmp_path, blob, with_full_names=True, name_prefix="efficientnetb0/", compress=True, block_entries=16)
    got = ci.import_savedmodel(path)
    assert got.dtype == np.float32 and np.array_equal(got, blob)
    info = ci.load_savedmodel(path)
    assert "dense_3/kernel" in info["named"] and info["reader"].tensor(info["named"]["dense_3/kernel"]).shape == (1024, 761)
    rd = info["reader"]
    assert rd.tensor("optimizer/iter/.ATTRIBUTES/VARIABLE_VALUE") == 12345
    small = info["named"]["stem_bn/gamma"]
    assert np.array_equal(rd.tensor(small, verify_crc=True), blob[[t for t in weights.manifest() if t["name"] == "stem_bn/gamma"][0]["offset"]:][:32])


def test_import_savedmodel_from_several_data_shards(tmp_path):
    """BundleHeaderProto.num_shards > 1: every entry names its shard; the reader opens variables.data-0000i-of-0000n lazily."""
    blob = weights.synthetic_blob(seed=4, calibrate=False)
    path = _write_model(tmp_path, blob, with_full_names=True, num_shards=3, block_entries=8)
    assert sorted(os.listdir(os.path.join(path, "variables"))) == ["variables.data-00000-of-00003", "variables.data-00001-of-00003",
                                                                     "variables.data-00002-of-00003", "variables.index"]
    assert np.array_equal(ci.import_savedmodel(path), blob)
    os.remove(os.path.join(path, "variables", "variables.data-00002-of-00003"))
    with pytest.raises((ci.CheckpointFormatError, OSError)):
        ci.import_savedmodel(path)


def test_import_savedmodel_positionally_when_names_are_missing(tmp_path):
    blob = weights.synthetic_blob(seed=4, calibrate=False)
    path = _write_model(tmp_path, blob, with_full_names=False)
    assert np.array_equal(ci.import_savedmodel(path), blob)


def test_import_errors(tmp_path):
    with pytest.raises(FileNotFoundError):
        ci.import_savedmodel(str(tmp_path))
    # a checkpoint of some other model: too few weighted layers, no matching names
    graph, keys = ub.keras_object_graph([("conv", ["kernel", "bias"])], with_full_names=True)
    os.makedirs(tmp_path / "m" / "variables")
    ub.write_bundle(str(tmp_path / "m" / "variables" / "variables"), {keys["conv/kernel"]: np.zeros((3, 3), np.float32),
                                                                      keys["conv/bias"]: np.zeros(3, np.float32)}, graph)
    with pytest.raises(ci.CheckpointFormatError):
        ci.import_savedmodel(str(tmp_path / "m"))
    (tmp_path / "bad").mkdir()
    (tmp_path / "bad" / "variables.index").write_bytes(b"\0" * 64)
    with pytest.raises(ci.CheckpointFormatError):
        ci.import_savedmodel(str(tmp_path / "bad" / "variables"))


def test_load_base_model_dispatches_on_a_savedmodel_directory(tmp_path, monkeypatch):
    """transfer_learning.load_base_model(path) takes what the reference's tf.keras.models.load_model takes."""
    from multilingual_kws_amd.embedding import transfer_learning as tl
    blob = weights.synthetic_blob(seed=5, calibrate=False)
    path = _write_model(tmp_path, blob, with_full_names=True)
    seen = {}

    class FakeEmbedding:
        def __init__(self, b, max_batch=0, output="dense_2"):
            seen["blob"] = b
    monkeypatch.setattr(tl, "EmbeddingModel", FakeEmbedding)
    _, got = tl.load_base_model(path, max_batch=4)
    assert np.array_equal(got, blob) and seen["blob"] is got


def test_hand_assembled_bundle_from_the_format_specs(golden_dir, tmp_path):
    """tests/golden/tf_bundle/: a tensor bundle written byte by byte from the LevelDB table-format / tensor_bundle.proto / snappy /
    CRC-32C specifications by tests/golden/make_tf_bundle_fixture.py, which shares no code with the reader or with util_bundle.py
    (one snappy-compressed data block: a literal of each length class and a 2-byte-offset copy; prefix-compressed keys; a
    two-entry restart array; masked block and tensor CRCs).  The first evidence for the reader that is not its author's writer."""
    import json
    import shutil
    d = os.path.join(golden_dir, "tf_bundle")
    E = json.load(open(os.path.join(d, "expected.json")))
    import hashlib
    assert hashlib.sha1(open(os.path.join(d, "variables.index"), "rb").read()).hexdigest() == E["index_sha1"]
    rd = ci.BundleReader(os.path.join(d, "variables"), verify=True)
    assert sorted(rd.entries) == sorted(E["tensors"]) and rd.num_shards == 1
    for key, exp in E["tensors"].items():
        t = rd.tensor(key, verify_crc=True)
        assert t.dtype == np.float32 and list(t.shape) == exp["shape"]
        assert np.array_equal(t.ravel(), np.asarray(exp["values"], np.float32))
        assert np.signbit(t.ravel()[1]) == np.signbit(np.float32(exp["values"][1]))          # -0.0 survives
        assert ci.crc32c(t.tobytes()) == exp["crc32c"] and rd.entries[key]["crc32c"] == ci.mask_crc(exp["crc32c"])
    # corruption is noticed at the right layer: a flipped tensor byte -> tensor CRC; a flipped index byte -> block CRC
    for name, offset, needs in (("variables.data-00000-of-00001", 14, "tensor checksum"), ("variables.index", 40, "block checksum")):
        bad = tmp_path / ("bad_" + name.split(".")[1][:4])
        shutil.copytree(d, bad)
        raw = bytearray(open(bad / name, "rb").read())
        raw[offset] ^= 0x10
        open(bad / name, "wb").write(bytes(raw))
        with pytest.raises(ci.CheckpointFormatError, match=needs):
            r2 = ci.BundleReader(str(bad / "variables"), verify=True)
            for key in r2.entries:
                r2.tensor(key, verify_crc=True)
    # the --verify report of tools/import_savedmodel.py: one line per tensor with shape, CRC verdict and sha1
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "import_savedmodel.py"), "--verify", d], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    for key, exp in E["tensors"].items():
        line = [ln for ln in out.stdout.splitlines() if ln.startswith(key + " ")]
        assert len(line) == 1 and exp["sha1"] in line[0] and "crc32c ok" in line[0] and str(exp["shape"]).replace(" ", "") in line[0].replace(" ", "")
    assert "2 tensors, 0 bad" in out.stdout


# ---- Keras .h5 (HDF5 without libhdf5) against files written by h5py: tests/golden/keras_h5/, tests/golden/make_h5_fixture.py ----------------
def test_h5_reader_against_h5py_written_files(golden_dir):
    """Every tensor of the two Keras-layout files (save_weights form and whole-model form) equals what the generator put in: 321
    datasets, ranks 0-4, float32 / float64 / int64 / big-endian, nested-model layer group, B-trees with several SNOD leaves,
    variable-length (h5py 3) / fixed-length / chunked (name0, name1) string attributes -- all written by h5py + libhdf5 1.10.6."""
    exp = np.load(os.path.join(golden_dir, "keras_h5", "expected.npz"))
    for fn in ("weights.h5", "model.h5"):
        r = ci.load_h5(os.path.join(golden_dir, "keras_h5", fn))
        named = r["named"]
        assert len(named) == len(exp.files) == 321
        for k in exp.files:
            got = named[k.replace("|", "/")]
            assert got.shape == exp[k].shape and got.dtype.kind == exp[k].dtype.kind and got.dtype.itemsize == exp[k].dtype.itemsize, k
            assert np.array_equal(got, exp[k]), (fn, k)
        assert [l for l, _ in r["layers"]] == ["efficientnetb0", "global_average_pooling2d", "dense", "dense_1", "dense_2"]     # layer_names order, not alphabetical
        assert r["layers"][0][1][:3] == ["normalization/mean:0", "normalization/variance:0", "normalization/count:0"] and len(r["layers"][0][1]) == 312
        assert r["layers"][3][1] == ["dense_1/kernel:0", "dense_1/bias:0"]            # joined from weight_names0 + weight_names1
    f = ci.H5File(os.path.join(golden_dir, "keras_h5", "model.h5"))
    attrs = f.attributes(f.root)
    assert attrs["keras_version"].item() == b"2.7.0" and attrs["backend"].item() == b"tensorflow" and b"Functional" in attrs["model_config"].item()
    root = f.members(f.root)
    assert set(root) == {"model_weights", "optimizer_weights", "compressed_extra"}
    assert int(f.dataset(dict(f.walk(root["optimizer_weights"]))["Adam/iter:0"])) == 12345
    with pytest.raises(ci.CheckpointFormatError, match="compressed|chunked"):        # refused loudly, never misread
        f.dataset(root["compressed_extra"])


def test_h5_errors(tmp_path, golden_dir):
    p = tmp_path / "not.h5"
    p.write_bytes(b"PK\x03\x04" + b"\x00" * 600)
    with pytest.raises(ci.CheckpointFormatError, match="signature"):
        ci.load_h5(str(p))
    raw = open(os.path.join(golden_dir, "keras_h5", "weights.h5"), "rb").read()
    q = tmp_path / "cut.h5"
    q.write_bytes(raw[:len(raw) // 2])
    with pytest.raises(ci.CheckpointFormatError, match="truncated"):
        ci.load_h5(str(q))
    v2 = bytearray(raw); v2[8] = 2
    (tmp_path / "v2.h5").write_bytes(bytes(v2))
    with pytest.raises(ci.CheckpointFormatError, match="superblock version 2"):
        ci.load_h5(str(tmp_path / "v2.h5"))
    with pytest.raises(ci.CheckpointFormatError, match="no variable|shape"):          # the reduced-extent fixture is not a full checkpoint
        ci.import_h5(os.path.join(golden_dir, "keras_h5", "weights.h5"))


def test_h5_corruption_fails_as_a_format_error(tmp_path, golden_dir):
    """A user block in front of the superblock (object addresses are relative to the base address, the end-of-file address is
    ABSOLUTE: `userblock.h5` was written by h5py with userblock_size=512, it is not patched together), truncation of such a file,
    and object / continuation / data addresses pointing outside the file: always CheckpointFormatError, never struct.error / IndexError."""
    import struct
    raw = open(os.path.join(golden_dir, "keras_h5", "weights.h5"), "rb").read()
    assert raw[:8] == ci.HDF5_SIGNATURE and raw[8] == 0
    ub_path = os.path.join(golden_dir, "keras_h5", "userblock.h5")
    blocked = open(ub_path, "rb").read()
    assert blocked[:8] != ci.HDF5_SIGNATURE and blocked[512:520] == ci.HDF5_SIGNATURE
    plain, moved = ci.H5File(os.path.join(golden_dir, "keras_h5", "weights.h5")), ci.H5File(ub_path)
    assert moved.base_addr == 512 and moved.eof == len(blocked)              # what libhdf5 wrote: eof is absolute
    assert list(moved.members(moved.root)) == list(plain.members(plain.root))
    exp = np.load(os.path.join(golden_dir, "keras_h5", "expected.npz"))
    got = ci.load_h5(ub_path)["named"]
    assert len(got) == len(exp.files)
    for k in exp.files:
        assert np.array_equal(got[k.replace("|", "/")], exp[k]), k
    cut = tmp_path / "userblock_cut.h5"
    cut.write_bytes(blocked[:-100])
    with pytest.raises(ci.CheckpointFormatError, match="truncated"):
        ci.H5File(str(cut))
    # addresses outside the file
    f = ci.H5File(os.path.join(golden_dir, "keras_h5", "weights.h5"))
    for bad in (len(raw) - 4, len(raw) + 1000, 1 << 40):
        with pytest.raises(ci.CheckpointFormatError):
            f.messages(bad)
        with pytest.raises(ci.CheckpointFormatError):
            f.dataset(bad)
        with pytest.raises(ci.CheckpointFormatError):
            f.members(bad)
    # a file whose root object-header address was overwritten, and one cut off behind the superblock with the end-of-file address patched
    broken = bytearray(raw)
    struct.pack_into("<Q", broken, 24 + 32 + 8, len(raw) + 12345)
    (tmp_path / "root.h5").write_bytes(bytes(broken))
    with pytest.raises(ci.CheckpointFormatError):
        ci.load_h5(str(tmp_path / "root.h5"))
    short = bytearray(raw[:len(raw) // 3])
    struct.pack_into("<Q", short, 24 + 16, len(short))
    (tmp_path / "short.h5").write_bytes(bytes(short))
    with pytest.raises(ci.CheckpointFormatError):
        ci.load_h5(str(tmp_path / "short.h5"))


H5PY_PYTHON = "/opt/conda/bin/python3.9"


@pytest.mark.skipif(not os.path.exists(H5PY_PYTHON), reason="no interpreter with h5py in this image")
def test_full_size_h5_checkpoint_round_trip(tmp_path, golden_dir, monkeypatch):
    """The whole 13 M-parameter blob written as a Keras whole-model .h5 by h5py at test time, imported back bit for bit, and
    load_base_model dispatching on the file name (the reference: tf.keras.models.load_model(base_model_path), transfer_learning.py:36)."""
    import subprocess
    from multilingual_kws_amd import weights
    from multilingual_kws_amd.embedding import transfer_learning as tl
    blob = weights.synthetic_blob(5, calibrate=False)
    np.save(tmp_path / "blob.npy", blob)
    manifest = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "embedding_manifest.json")
    out = str(tmp_path / "multilingual_context_73_0.8011.h5")
    subprocess.run([H5PY_PYTHON, os.path.join(golden_dir, "make_h5_fixture.py"), "--full", out, str(tmp_path / "blob.npy"), manifest], check=True,
                   env={k: v for k, v in os.environ.items() if not k.startswith("PYTHON")})
    assert os.path.getsize(out) > 50e6
    got = ci.import_h5(out)
    assert got.dtype == np.float32 and np.array_equal(got, blob)
    seen = {}
    monkeypatch.setattr(tl, "EmbeddingModel", lambda b, max_batch, output: seen.setdefault("blob", b) is None or "handle")
    tl.load_base_model(out, max_batch=4)
    assert np.array_equal(seen["blob"], blob)
