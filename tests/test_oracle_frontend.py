"""Pins oracle/microfrontend_oracle.c to upstream TensorFlow's micro-frontend unit-test constants
and to the SURVEY.md Appendix D checksums (tests/golden/frontend_golden.json)."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle.frontend_oracle import FrontendOracle
from tests.util_signals import d3_inputs, read_wav_pcm16


@pytest.fixture(scope="module")
def G(golden_dir):
    return json.load(open(os.path.join(golden_dir, "frontend_golden.json")))


def _small(G, **over):
    cfg = dict(G["upstream_tf"]["config"])
    cfg.update(over)
    return FrontendOracle(**cfg)


def test_upstream_end_to_end_known_answer(G):
    u = G["upstream_tf"]
    pcm = np.array(u["audio_pattern"] * u["audio_repeats"], dtype=np.int16)
    assert _small(G).run_i16(pcm).tolist() == u["frontend_output"]
    assert _small(G, enable_pcan=False).run_i16(pcm).tolist() == G["derived_small_config"]["pcan_off_output"]


def test_upstream_window_stage(G):
    u = G["upstream_tf"]
    fo = _small(G)
    assert fo.table("window_coef").tolist() == u["window_coefficients"]
    pcm = np.array(u["audio_pattern"] * u["audio_repeats"], dtype=np.int16)
    w, m = fo.window_frame(pcm[:25])
    assert w.tolist() == u["windowed_frame0"] and m == u["window_max_abs"]


def test_upstream_noise_reduction_stage(G):
    u = G["upstream_tf"]
    est, sig = _small(G).noise_reduction([0, 0], u["noise_reduction_input"])
    assert est.tolist() == u["noise_reduction_estimate"]
    assert sig.tolist() == u["noise_reduction_output"]


def test_upstream_filterbank_tables(G):
    fb = G["upstream_tf"]["filterbank"]
    fo = _small(G)
    assert fo.scalar("start_index") == fb["start_index"] and fo.scalar("end_index") == fb["end_index"]
    assert fo.table("chan_freq_starts").tolist() == fb["freq_starts"]
    assert fo.table("chan_widths").tolist() == fb["widths"]
    assert fo.table("weights").tolist() == fb["weights"]


def test_log_lut_and_log(G):
    u = G["upstream_tf"]
    fo = FrontendOracle()
    lut = fo.table("log_lut")
    assert lut[:12].tolist() == u["log_lut_head"]
    assert [int(lut.argmax()), int(lut.max())] == u["log_lut_peak"]
    assert lut[-4:].tolist() == u["log_lut_tail"]
    assert fo.log(3578 << 3) == G["derived_small_config"]["log_3578_shl3"]
    assert fo.log(1533 << 3) == G["derived_small_config"]["log_1533_shl3"]


def test_real_config_tables(G):
    s = G["survey"]
    fo = FrontendOracle()
    coef = fo.table("window_coef")
    assert hashlib.sha1(coef.astype("<i2").tobytes()).hexdigest() == s["window_coef_sha1"]
    assert int(coef.astype(np.int64).sum()) == s["window_coef_sum"]
    W, U = fo.table("weights"), fo.table("unweights")
    assert len(W) == s["filterbank_num_weights"]
    assert hashlib.sha1(W.astype("<i2").tobytes() + U.astype("<i2").tobytes()).hexdigest() == s["filterbank_sha1_W_then_U"]
    assert int(W.astype(np.int64).sum()) == s["filterbank_sum_W"] and int(U.astype(np.int64).sum()) == s["filterbank_sum_U"]
    assert fo.scalar("start_index") == s["start_index"] and fo.scalar("end_index") == s["end_index"]
    assert fo.table("chan_freq_starts").tolist() == s["aligned_freq_starts"]
    lut = fo.table("gain_lut")
    assert lut[:2].tolist() == s["pcan_lut_head"] and lut[2::4].tolist() == s["pcan_lut_y0"]


@pytest.mark.parametrize("name", ["square4", "lcg", "sine1k", "zeros"])
@pytest.mark.parametrize("pcan", ["on", "off"])
def test_real_config_checksums(G, name, pcan):
    sha, total, mx, head = G["survey"]["real_config_outputs"][name][pcan]
    out = FrontendOracle(enable_pcan=(pcan == "on")).run_i16(d3_inputs()[name])
    assert out.shape == (49, 40)
    assert hashlib.sha1(out.astype("<u2").tobytes()).hexdigest()[:16] == sha
    assert int(out.sum()) == total and int(out.max()) == mx and out[0, :10].tolist() == head


@pytest.mark.parametrize("clip", [0, 1, 2])
def test_real_speech_fixtures(G, golden_dir, clip):
    info = G["survey"]["real_speech"][f"clip{clip}"]
    pcm, raw = read_wav_pcm16(os.path.join(golden_dir, f"tutorial_clip{clip}.wav"))
    assert hashlib.sha1(raw).hexdigest()[:16] == info["wav_sha1"]
    npz = np.load(os.path.join(golden_dir, "frontend_real_speech.npz"))
    for pcan in ("on", "off"):
        out = FrontendOracle(enable_pcan=(pcan == "on")).run_i16(pcm)
        sha, total, mx = info[pcan]
        assert hashlib.sha1(out.astype("<u2").tobytes()).hexdigest()[:16] == sha
        assert int(out.sum()) == total and int(out.max()) == mx
        assert np.array_equal(out, npz[f"clip{clip}_pcan_{pcan}"])


def test_batch_float_path_matches_int_path():
    """to_micro_spectrogram semantics: float * 32768 -> trunc -> int16; output = raw * 10/256."""
    rng = np.random.default_rng(0)
    pcm = rng.integers(-20000, 20000, size=(3, 16000)).astype(np.int16)
    fo = FrontendOracle()
    f32, u16 = fo.run_batch_f32(pcm.astype(np.float32) / 32768.0, want_u16=True)
    for b in range(3):
        assert np.array_equal(u16[b], fo.run_i16(pcm[b]))
    assert np.array_equal(f32, u16.astype(np.float32) * np.float32(10.0 / 256.0))
    k = f32 / 0.0390625
    assert np.array_equal(k, np.round(k))


def test_edge_cases():
    fo = FrontendOracle()
    assert fo.num_frames(479) == 0 and fo.num_frames(480) == 1 and fo.num_frames(16000) == 49
    assert fo.run_i16(np.zeros(100, dtype=np.int16)).shape == (0, 40)
    full = np.full(16000, -32768, dtype=np.int16)   # |-32768| stays negative in int16: never wins max_abs
    out = fo.run_i16(full)
    assert out.shape == (49, 40) and out.max() < 700
    # saturating float->int16 (SURVEY risk R4): +1.0 -> 32767, -1.0 -> -32768
    a = np.array([[1.0] * 16000, [-1.0] * 16000], dtype=np.float32)
    _, u16 = fo.run_batch_f32(a, want_u16=True)
    assert np.array_equal(u16[0], fo.run_i16(np.full(16000, 32767, dtype=np.int16)))
    assert np.array_equal(u16[1], out)
    # sqrt64 rounding rule incl. the 32-bit path's 0xFFFF saturation
    for x in [0, 1, 2, 3, 4, 6, 7, 2**32 - 1, 4294901761, 2**32, 2**40 + 12345, 2**63, 2**64 - 1]:
        r = int(np.floor(np.sqrt(float(x))))
        while r * r > x:
            r -= 1
        while (r + 1) * (r + 1) <= x:
            r += 1
        exp = r + 1 if (x - r * r > r) else r
        if x < 2**32 and r == 0xFFFF:
            exp = r
        if r == 0xFFFFFFFF:
            exp = r
        assert fo.sqrt64(x) == exp, x
