"""The frontend oracle -- and, `-m gpu`, the HIP frontend -- held to an OUTPUT OF THE REFERENCE ITSELF.

`/root/reference/multilingual_kws_intro_tutorial.ipynb` cell 13 renders `input_data.file2spec(settings, clip)`
(`input_data.py:38-47` -> the real TF `AudioMicrofrontend` op of `:19-35`) for three clips with `imshow`; the
clips' WAV bytes are embedded in cell 11.  `tests/golden/make_tutorial_png_golden.py` (build container) asserts that
correspondence and commits the PNG (`tutorial_cell13.png`) + matplotlib's viridis byte table.  Here the panels and
their 49 x 40 cell grids are detected from the pixels (`tests/util_png.py`) and every cell's COLOUR must be
`viridis[floor(256 (x - min) / (max - min))]` of what we compute for the same clip.

What this pins, honestly: the whole integer pipeline end to end (window, FFT, filterbank, noise reduction, PCAN,
log, `x 10/256`) on real speech, with the op's defaults as the reference runs it -- at the resolution of the
rendering: one colour step = (max - min) / 256 ~ 0.1 feature units ~ 2.5 raw integer units (of up to ~650).  Bit-exactness
below that step still rests on upstream TF's unit-test constants (`test_oracle_frontend.py`).  The negative cases show
the resolving power: every op default the reference leaves implicit, set wrong, recolours tens to thousands of cells.
"""
import os

import numpy as np
import pytest

from tests.util_png import decode_png, imshow_expected_index, imshow_panels, index_distance, load_viridis
from tests.util_signals import read_wav_pcm16

SHAPE = (49, 40)


@pytest.fixture(scope="module")
def panels(golden_dir):
    lut = load_viridis(golden_dir)
    img = decode_png(open(os.path.join(golden_dir, "tutorial_cell13.png"), "rb").read())
    assert img.shape == (248, 592, 4)
    p = imshow_panels(img, lut, SHAPE)
    assert len(p) == 3
    return lut, p


def _clip(golden_dir, i):
    return read_wav_pcm16(os.path.join(golden_dir, f"tutorial_clip{i}.wav"))[0]


def _mismatch(lut, cells, feats):
    return index_distance(cells, lut, imshow_expected_index(feats))


def test_png_decoder_against_pillow(golden_dir):
    PIL = pytest.importorskip("PIL.Image")
    path = os.path.join(golden_dir, "tutorial_cell13.png")
    assert np.array_equal(decode_png(open(path, "rb").read()), np.array(PIL.open(path)))


def test_every_pixel_of_a_panel_is_a_colormap_colour_and_cells_are_blocks(panels):
    lut, p = panels                                   # imshow_panels asserts both while it reads the cells
    for cells in p:
        assert cells.shape == SHAPE + (3,)
        assert len({tuple(c) for c in cells.reshape(-1, 3)}) > 100      # real content, not a flat image


@pytest.mark.parametrize("clip", [0, 1, 2])
def test_oracle_reproduces_the_reference_rendering(panels, golden_dir, clip):
    """PCAN on (the op's default, SURVEY risk R1): ALL 1 960 cells of each clip carry exactly the reference's colour."""
    from oracle.frontend_oracle import FrontendOracle
    lut, p = panels
    feats = FrontendOracle().run_batch_f32(_clip(golden_dir, clip).astype(np.float32) / np.float32(32768.0))[0]
    d = _mismatch(lut, p[clip], feats)
    assert (d == 0).mean() >= 0.995 and d.max() <= 1, ((d != 0).sum(), d.max())
    assert (d == 0).all()                             # measured: 5 880 of 5 880


# every default of the op that `input_data.py:25-33` does NOT pass, set to a plausible wrong value: the rendering must
# tell.  Thresholds are a third of the measured number of recoloured cells (over the three clips).
WRONG = [
    ("enable_pcan", dict(enable_pcan=False), 1000),            # measured 3 208 of 5 880 cells (risk R1: settled)
    ("enable_log", dict(enable_log=False), 700),               # 2 084
    ("pcan_strength", dict(pcan_strength=0.5), 800),           # 2 381
    ("pcan_offset", dict(pcan_offset=40.0), 500),              # 1 635
    ("scale_shift", dict(scale_shift=5), 200),                 # 596
    ("even_smoothing", dict(even_smoothing=0.06), 700),        # 2 060
    ("odd_smoothing", dict(odd_smoothing=0.025), 600),         # 1 805
    ("upper_band_limit", dict(upper_band_limit=7999.0), 650),  # 1 973
    ("lower_band_limit", dict(lower_band_limit=20.0), 700),    # 2 085
    ("window 25 ms", dict(window_size_ms=25), 600),            # 1 925
    ("gain_bits", dict(gain_bits=20), 9),                      # 28: a uniform gain mostly cancels in min/max normalisation
    ("smoothing_bits", dict(smoothing_bits=9), 2),             # 8: only the noise estimate's rounding moves
]


@pytest.mark.parametrize("name,over,least", WRONG, ids=[w[0] for w in WRONG])
def test_wrong_defaults_are_visible_in_the_rendering(panels, golden_dir, name, over, least):
    from oracle.frontend_oracle import FrontendOracle
    lut, p = panels
    fo, bad = FrontendOracle(**over), 0
    for clip in range(3):
        feats = fo.run_batch_f32(_clip(golden_dir, clip).astype(np.float32) / np.float32(32768.0))[0]
        bad += int((_mismatch(lut, p[clip], feats) != 0).sum())
    assert bad >= least, (name, bad)


def test_what_the_rendering_cannot_see(panels, golden_dir):
    """Recorded so that nobody reads more into the pin than it holds: `min_signal_remaining` 0.05 -> 0.1 changes no cell's
    colour on these clips (speech sits far above the noise floor)."""
    from oracle.frontend_oracle import FrontendOracle
    lut, p = panels
    fo = FrontendOracle(min_signal_remaining=0.1)
    for clip in range(3):
        feats = fo.run_batch_f32(_clip(golden_dir, clip).astype(np.float32) / np.float32(32768.0))[0]
        assert (_mismatch(lut, p[clip], feats) == 0).all()


def test_truncating_cast_is_what_the_reference_does(panels, golden_dir):
    """`tf.cast(audio * 32768, int16)` on decoded PCM is the identity; a frontend fed the float path must agree with
    the int16 path on these clips (this is the path `file2spec` takes)."""
    from oracle.frontend_oracle import FrontendOracle
    fo = FrontendOracle()
    for clip in range(3):
        pcm = _clip(golden_dir, clip)
        f32, u16 = fo.run_batch_f32(pcm.astype(np.float32) / np.float32(32768.0), want_u16=True)
        assert np.array_equal(u16[0], fo.run_i16(pcm))


@pytest.mark.gpu
@pytest.mark.parametrize("pcan", [1, 0])
def test_hip_frontend_reproduces_the_reference_rendering(panels, golden_dir, pcan):
    torch = pytest.importorskip("torch")
    from multilingual_kws_amd.frontend import Frontend
    lut, p = panels
    pcm = np.stack([_clip(golden_dir, i) for i in range(3)])
    audio = torch.from_numpy(pcm.astype(np.float32) / np.float32(32768.0)).to("cuda:0")
    feats = Frontend(enable_pcan=pcan).forward(audio).cpu().numpy()
    bad = [int((_mismatch(lut, p[i], feats[i]) != 0).sum()) for i in range(3)]
    if pcan:
        assert bad == [0, 0, 0], bad
    else:
        assert min(bad) > 500, bad                    # the negative: PCAN off is NOT what the reference ran


@pytest.mark.gpu
def test_dropin_file2spec_reproduces_the_reference_rendering(panels, golden_dir):
    """The reference's own call, verbatim, on the drop-in's import path (cell 13)."""
    from multilingual_kws.embedding import input_data
    lut, p = panels
    settings = input_data.standard_microspeech_model_settings(label_count=1)
    for i in range(3):
        spectrogram = input_data.file2spec(settings, os.path.join(golden_dir, f"tutorial_clip{i}.wav"))
        spectrogram = spectrogram.cpu().numpy() if hasattr(spectrogram, "cpu") else np.asarray(spectrogram)
        assert spectrogram.shape == SHAPE
        assert (_mismatch(lut, p[i], spectrogram) == 0).all()
