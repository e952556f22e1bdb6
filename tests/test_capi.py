"""C-ABI checks that need no GPU: the library loads, exports every symbol include/mkws.h declares,
builds the same integer tables as the oracle, and refuses to run without a device."""
import ctypes
import hashlib
import json
import os
import re

import numpy as np
import pytest

from multilingual_kws_amd import _lib, frontend, weights
from oracle import efficientnet_oracle as eo
from oracle.frontend_oracle import FrontendOracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "mkws.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mkws_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    L = ctypes.CDLL(_lib.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(L, name), f"{name} is declared in include/mkws.h but not exported"
    bound = {n for n, _, _ in _lib.SYMBOLS}
    assert set(declared) == bound, (set(declared) ^ bound)
    lib = _lib.lib()
    assert lib.mkws_abi_version() == _lib.ABI_VERSION == 5 and lib.mkws_build_arch() == b"gfx950"


CONFIGS = [
    {},
    dict(enable_pcan=0),
    dict(sample_rate=1000, window_size_ms=25, window_step_ms=10, num_channels=2, upper_band_limit=450.0, lower_band_limit=8.0),
    dict(sample_rate=8000, window_size_ms=40, window_step_ms=20, num_channels=32, upper_band_limit=3800.0, lower_band_limit=60.0),
    dict(window_size_ms=25, window_step_ms=10, num_channels=32),
    dict(sample_rate=16000, num_channels=10, lower_band_limit=20.0, upper_band_limit=4000.0, smoothing_bits=12, gain_bits=20),
]


@pytest.mark.parametrize("over", CONFIGS)
def test_host_tables_match_oracle(over):
    cfg = frontend.make_cfg(**over)
    fo = FrontendOracle(**{k: (bool(v) if k == "enable_pcan" else v) for k, v in over.items()})
    sc = frontend.host_scalars(cfg)
    for k in ("window_size", "window_step", "fft_size", "start_index", "end_index", "num_weights", "correction_bits"):
        assert sc[k] == fo.scalar(k), k
    for name in ("window_coef", "twiddles", "super_twiddles", "weights", "unweights", "chan_freq_starts",
                 "chan_weight_starts", "chan_widths", "log_lut"):
        assert np.array_equal(frontend.host_table(cfg, name), fo.table(name)), name
    if over.get("enable_pcan", 1):
        assert sc["snr_shift"] == fo.scalar("snr_shift")
        assert np.array_equal(frontend.host_table(cfg, "gain_lut"), fo.table("gain_lut"))


def test_host_tables_match_golden_checksums(golden_dir):
    s = json.load(open(os.path.join(golden_dir, "frontend_golden.json")))["survey"]
    cfg = frontend.make_cfg()
    assert hashlib.sha1(frontend.host_table(cfg, "window_coef").astype("<i2").tobytes()).hexdigest() == s["window_coef_sha1"]
    W, U = frontend.host_table(cfg, "weights"), frontend.host_table(cfg, "unweights")
    assert hashlib.sha1(W.astype("<i2").tobytes() + U.astype("<i2").tobytes()).hexdigest() == s["filterbank_sha1_W_then_U"]
    assert frontend.host_table(cfg, "gain_lut")[2::4].tolist() == s["pcan_lut_y0"]


def test_num_frames_and_bad_configs():
    cfg = frontend.make_cfg()
    assert [frontend.num_frames(cfg, n) for n in (0, 479, 480, 799, 800, 16000)] == [0, 0, 1, 1, 2, 49]
    bad = frontend.make_cfg(upper_band_limit=9000.0)     # above Nyquist -> filterbank end_index error upstream too
    with pytest.raises(_lib.MkwsError):
        frontend.host_table(bad, "weights")
    with pytest.raises(_lib.MkwsError):
        frontend.host_table(frontend.make_cfg(num_channels=0), "weights")
    with pytest.raises(TypeError):
        frontend.make_cfg(not_an_option=1)


def test_weight_manifest_matches_oracle_architecture():
    man = weights.manifest()
    assert [(t["name"], tuple(t["shape"])) for t in man] == eo.tensor_list()
    assert weights.weight_count() == eo.blob_size() == 12967004 + 2      # Keras params to dense_2 + Normalization mean/var
    offs = [t["offset"] for t in man]
    assert offs == sorted(offs) and offs[0] == 0
    assert all(a["offset"] + a["count"] == b["offset"] for a, b in zip(man, man[1:]))


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    L = _lib.lib()
    h = ctypes.c_void_p()
    cfg = frontend.make_cfg()
    assert L.mkws_frontend_create(ctypes.byref(cfg), 16000, ctypes.byref(h)) == -3      # MKWS_ERR_NO_DEVICE
    assert b"no HIP device" in L.mkws_last_error() or b"fallback" in L.mkws_last_error()
    blob = np.zeros(weights.weight_count(), dtype=np.float32)
    assert L.mkws_embed_create(blob.ctypes.data, blob.shape[0], 4, ctypes.byref(h)) == -3
    assert L.mkws_head_create(1024, 18, 3, 4, ctypes.byref(h)) == -3
    assert L.mkws_embed_create(blob.ctypes.data, 5, 4, ctypes.byref(h)) == -6         # MKWS_ERR_BAD_WEIGHTS comes first
