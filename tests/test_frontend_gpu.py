"""-m gpu: the HIP micro-frontend against the oracle (bit-exact) and the golden fixtures, through the C-ABI."""
import hashlib
import json
import os

import numpy as np
import pytest

from tests.util_signals import d3_inputs, read_wav_pcm16

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _oracle(**kw):
    from oracle.frontend_oracle import FrontendOracle
    return FrontendOracle(**kw)


def _fe(**kw):
    from multilingual_kws_amd.frontend import Frontend
    return Frontend(**kw)


def _u16(t):
    return t.cpu().numpy().view(np.uint16)


@pytest.mark.parametrize("pcan", [1, 0])
def test_golden_checksums_on_device(dev, golden_dir, pcan):
    """SURVEY Appendix D.3/D.4 checksums reproduced by the GPU kernel itself."""
    G = json.load(open(os.path.join(golden_dir, "frontend_golden.json")))["survey"]
    fe = _fe(enable_pcan=pcan)
    key = "on" if pcan else "off"
    names = list(d3_inputs())
    pcm = np.stack([d3_inputs()[n] for n in names] + [read_wav_pcm16(os.path.join(golden_dir, f"tutorial_clip{i}.wav"))[0] for i in range(3)])
    _, raw = fe.forward(torch.from_numpy(pcm).to(dev), want_raw=True)
    raw = _u16(raw)
    for j, n in enumerate(names):
        sha, total, mx, head = G["real_config_outputs"][n][key]
        assert hashlib.sha1(raw[j].astype("<u2").tobytes()).hexdigest()[:16] == sha
        assert int(raw[j].sum()) == total and int(raw[j].max()) == mx and raw[j][0, :10].tolist() == head
    npz = np.load(os.path.join(golden_dir, "frontend_real_speech.npz"))
    for i in range(3):
        sha, total, mx = G["real_speech"][f"clip{i}"][key]
        assert hashlib.sha1(raw[4 + i].astype("<u2").tobytes()).hexdigest()[:16] == sha
        assert np.array_equal(raw[4 + i], npz[f"clip{i}_pcan_{key}"])


@pytest.mark.parametrize("pcan", [1, 0])
def test_random_and_extreme_clips_bit_exact(dev, pcan):
    rng = np.random.default_rng(10 + pcan)
    clips = [rng.integers(-a, a + 1, size=16000).astype(np.int16) for a in (1, 3, 30, 300, 3000, 12000, 30000, 32767)]
    clips += [np.full(16000, v, dtype=np.int16) for v in (-32768, 32767, 1, -1)]
    ramp = (np.arange(16000) * 4 % 65536 - 32768).astype(np.int16)
    imp = np.zeros(16000, dtype=np.int16); imp[::997] = 32767; imp[5::1201] = -32768
    clips += [ramp, imp]
    pcm = np.stack(clips)
    fo = _oracle(enable_pcan=bool(pcan))
    exp = np.stack([fo.run_i16(c) for c in pcm])
    fe = _fe(enable_pcan=pcan)
    spec, raw = fe.forward(torch.from_numpy(pcm).to(dev), want_raw=True)
    assert np.array_equal(_u16(raw), exp)
    assert np.array_equal(spec.cpu().numpy(), exp.astype(np.float32) * np.float32(10 / 256))
    # float input path == int16 path == oracle float path (to_micro_spectrogram semantics)
    f = torch.from_numpy(pcm.astype(np.float32) / 32768).to(dev)
    spec_f, raw_f = fe.forward(f, want_raw=True)
    assert torch.equal(raw_f, raw) and torch.equal(spec_f, spec)
    assert np.array_equal(spec_f.cpu().numpy(), fo.run_batch_f32(pcm.astype(np.float32) / 32768))


def test_float_cast_semantics_and_saturation(dev):
    """audio*32768 truncates toward zero; +-1.0 saturate."""
    rng = np.random.default_rng(3)
    a = rng.uniform(-1, 1, size=(6, 16000)).astype(np.float32)
    a[4] = 1.0
    a[5] = -1.0
    a[3, ::7] = 0.99999
    fe, fo = _fe(), _oracle()
    spec, raw = fe.forward(torch.from_numpy(a).to(dev), want_raw=True)
    e_spec, e_raw = fo.run_batch_f32(a, want_u16=True)
    assert np.array_equal(_u16(raw), e_raw) and np.array_equal(spec.cpu().numpy(), e_spec)


def test_shapes_empty_short_and_unaligned(dev):
    fe = _fe(max_samples=20000)
    assert fe.forward(torch.zeros((0, 16000), device=dev)).shape == (0, 49, 40)          # empty batch
    assert fe.forward(torch.zeros((3, 479), device=dev)).shape == (3, 0, 40)             # shorter than one window
    fo = _oracle()
    rng = np.random.default_rng(4)
    for n in (480, 799, 801, 4001, 15999, 16001, 20000):                                  # odd lengths: unaligned load path
        a = rng.uniform(-0.9, 0.9, size=(5, n)).astype(np.float32)
        spec = fe.forward(torch.from_numpy(a).to(dev))
        assert spec.shape == (5, fo.num_frames(n), 40)
        assert np.array_equal(spec.cpu().numpy(), fo.run_batch_f32(a)), n
    one = fe.forward(torch.from_numpy(a[0]).to(dev))                                      # 1-D input like the reference op
    assert one.shape == (1, fo.num_frames(20000), 40)
    from multilingual_kws_amd._lib import MkwsError
    with pytest.raises(MkwsError):
        fe.forward(torch.zeros((1, 20001), device=dev))                                   # beyond max_samples
    with pytest.raises(MkwsError):
        _fe(window_size_ms=10)                                                            # 160-sample window: FFT 256 unsupported


def test_full_batch_properties(dev):
    """BASELINE size (B=1024): EVERY clip bit for bit against the C oracle (raw integers and scaled features), clip independence / permutation
    invariance on all."""
    from multilingual_kws_amd import synth
    a = synth.clips_float32(1024)
    fe = _fe()
    x = torch.from_numpy(a).to(dev)
    spec, raw = fe.forward(x, want_raw=True)
    ref_spec, ref_raw = _oracle().run_batch_f32(a, want_u16=True)
    assert np.array_equal(spec.cpu().numpy(), ref_spec)
    assert np.array_equal(raw.cpu().numpy().view(np.uint16), ref_raw)
    perm = torch.randperm(1024, device=dev)
    assert torch.equal(fe.forward(x[perm]), spec[perm])                                   # clips are independent
    assert torch.equal(fe.forward(x[5:6])[0], spec[5])                                     # batch-size invariant
    k = spec / 0.0390625
    assert torch.equal(k, torch.round(k)) and float(spec.max()) < 27.0                    # exact multiples of 10/256
    assert torch.equal(fe.forward(x), spec)                                                # deterministic


def test_other_configurations(dev):
    rng = np.random.default_rng(8)
    a = rng.uniform(-0.7, 0.7, size=(4, 16000)).astype(np.float32)
    for kw in (dict(window_size_ms=25, window_step_ms=10, num_channels=32),
               dict(num_channels=10, lower_band_limit=20.0, upper_band_limit=4000.0, smoothing_bits=12, gain_bits=20),
               dict(enable_log=0), dict(scale_shift=4, even_smoothing=0.1, odd_smoothing=0.2, min_signal_remaining=0.2)):
        fe = _fe(**kw)
        fo = _oracle(**{k: (bool(v) if k.startswith("enable") else v) for k, v in kw.items()})
        _, raw = fe.forward(torch.from_numpy(a).to(dev), want_raw=True)
        assert np.array_equal(_u16(raw), fo.run_batch_f32(a, want_u16=True)[1]), kw


def test_streaming_shares_frames_bit_exactly(dev):
    """batch_streaming_analysis.py:99-117: a window every 320 samples == the op on each 1 s slice."""
    rng = np.random.default_rng(9)
    n = 16000 * 4 + 123
    s = rng.uniform(-0.8, 0.8, size=n).astype(np.float32)
    fe = _fe(max_samples=n)
    x = torch.from_numpy(s).to(dev)
    for hop in (320, 640, 1600):
        sp, raw = fe.stream(x, 16000, hop, want_raw=True)
        nw = 1 + (n - 16000) // hop
        assert sp.shape == (nw, 49, 40)
        wins = np.stack([s[w * hop:w * hop + 16000] for w in range(nw)])
        assert np.array_equal(sp.cpu().numpy(), _oracle().run_batch_f32(wins))
    assert fe.stream(x[:15999], 16000, 320).shape == (0, 49, 40)
    from multilingual_kws_amd._lib import MkwsError
    with pytest.raises(MkwsError):
        fe.stream(x, 16000, 300)                                                          # hop not a multiple of the frame step


def test_to_micro_spectrogram_api(dev):
    from multilingual_kws_amd.embedding import input_data
    ms = input_data.standard_microspeech_model_settings(3)
    rng = np.random.default_rng(11)
    a = rng.uniform(-0.9, 0.9, size=16000).astype(np.float32)
    out = input_data.to_micro_spectrogram(ms, a)                                          # numpy in -> numpy out, [49,40]
    assert isinstance(out, np.ndarray) and out.shape == (49, 40)
    assert np.array_equal(out, _oracle().run_batch_f32(a)[0])
    t = input_data.to_micro_spectrogram(ms, torch.from_numpy(np.stack([a, -a])).to(dev))
    assert t.is_cuda and t.shape == (2, 49, 40) and np.array_equal(t[0].cpu().numpy(), out)
