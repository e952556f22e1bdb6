"""Host-side logic of the drop-in Python surface (no GPU needed)."""
import hashlib
import os

import numpy as np
import pytest

from multilingual_kws_amd import arch, synth
from multilingual_kws_amd.embedding import input_data, transfer_learning
from tests.util_data import make_fewshot_dataset, wav_bytes, write_wav


def test_model_settings_keys_and_values():
    ms = input_data.standard_microspeech_model_settings(3)
    assert ms == {"desired_samples": 16000, "window_size_samples": 480, "window_stride_samples": 320,
                  "spectrogram_length": 49, "fingerprint_width": 40, "fingerprint_size": 1960, "label_count": 3,
                  "sample_rate": 16000, "preprocess": "micro", "average_window_width": -1}
    avg = input_data.prepare_model_settings(12, 16000, 1000, 30, 10, 40, "average")
    assert avg["spectrogram_length"] == 98 and avg["average_window_width"] == 6 and avg["fingerprint_width"] == 43
    assert input_data.prepare_model_settings(2, 16000, 20, 30, 10, 40, "mfcc")["spectrogram_length"] == 0
    with pytest.raises(ValueError):
        input_data.prepare_model_settings(2, 16000, 1000, 30, 20, 40, "nope")
    assert (input_data.SILENCE_LABEL, input_data.SILENCE_INDEX, input_data.UNKNOWN_WORD_LABEL, input_data.UNKNOWN_WORD_INDEX) == \
        ("_silence_", 0, "_unknown_", 1)
    assert [input_data._next_power_of_two(x) for x in (0, 1, 2, 3, 480, 512, 513)] == [1, 1, 2, 4, 512, 512, 1024]


def test_decode_wav_semantics():
    pcm = np.array([0, 16384, -16384, 32767, -32768], dtype=np.int16)
    x, rate = input_data.decode_wav(wav_bytes(pcm), desired_samples=8)
    assert rate == 16000 and x.dtype == np.float32
    assert np.array_equal(x, np.array([0, .5, -.5, 32767 / 32768, -1, 0, 0, 0], dtype=np.float32))     # zero-padded
    assert np.array_equal(input_data.decode_wav(wav_bytes(pcm), desired_samples=3)[0], x[:3])           # truncated
    assert input_data.decode_wav(wav_bytes(pcm))[0].shape == (5,)
    stereo = np.stack([pcm, -pcm // 2], axis=1).reshape(-1)
    assert np.array_equal(input_data.decode_wav(wav_bytes(stereo, channels=2), 5)[0], x[:5])            # first channel kept
    with pytest.raises(ValueError):
        input_data.decode_wav(b"not a wav file at all")
    # extra chunk before "data" is skipped
    w = wav_bytes(pcm)
    w2 = w[:36] + b"LIST" + (4).to_bytes(4, "little") + b"abcd" + w[36:]
    assert np.array_equal(input_data.decode_wav(w2, 5)[0], x[:5])


def test_add_background_numpy():
    fg = np.array([0.5, -0.5, 0.5, -0.5], dtype=np.float32)
    bg = np.array([0.1, 0.1, -0.1, -0.1], dtype=np.float32)
    out = input_data.add_background(fg, bg, 0.5)            # rms ratio 5 -> bg*5*0.5 + fg
    assert np.allclose(out, fg + bg * 2.5)
    assert np.allclose(input_data.add_background(fg, np.zeros(4, np.float32), 1.0), fg)    # silent background: scaling 0
    assert input_data.add_background(fg * 2, bg * 30, 1.0).max() <= 1.0                     # clipped


def test_audio_dataset_label_order(tmp_path):
    d = make_fewshot_dataset(str(tmp_path), n_train=2, n_val=1, n_unknown=3, n_bg=1)
    ms = input_data.standard_microspeech_model_settings(3)
    ds = input_data.AudioDataset(ms, ["tiempo"], d["bg_dir"], d["unknown"], unknown_percentage=50.0, seed=1)
    assert ds.commands == ["_silence_", "_unknown_", "tiempo"]
    assert ds.max_time_shift_samples == 1600
    assert ds._label_id("tiempo") == 2 and ds._label_id("_unknown_") == 1 and ds._label_id("nope") == 0
    assert ds.get_label(os.path.join("a", "word", "x.wav")) == "word"
    assert ds.background_sizes.tolist() == [96000] and ds.background_host.shape == (1, 96000)
    assert input_data.AudioDataset(ms, ["w"], d["bg_dir"], [], silence_percentage=0).commands == ["w"]
    assert input_data.AudioDataset(ms, ["w"], d["bg_dir"], d["unknown"], unknown_percentage=0).commands == ["_silence_", "w"]
    train = ds.init_single_target(input_data.AUTOTUNE, d["train"], is_training=True)
    assert len(train) == 2 and train.labels == ["tiempo", "tiempo"]
    ev = ds.eval_with_silence_unknown(input_data.AUTOTUNE, d["train"] * 10, label_from_parent_dir=False)
    assert (ev.extra_silence, ev.extra_unknown) == (2, 10)
    par = ds.init_from_parent_dir(input_data.AUTOTUNE, d["unknown"], is_training=False)
    assert par.labels == ["other"] * 3
    # vectorised draws follow the reference's ranges
    m = ds._draw_specaug_masks(4000)
    assert m[:, [1, 3, 5, 7]].max() <= 2 and (m[:, 0] + m[:, 1]).max() <= 40 and (m[:, 4] + m[:, 5]).max() <= 49
    frac_any = (m[:, [1, 3, 5, 7]].sum(1) > 0).mean()
    assert 0.6 < frac_any < 0.8          # 80 % apply x (1 - P[freq_n = 0 and time_n = 0] = 8/9)
    sh = ds._draw_shift(5000)
    assert sh.min() >= -1600 and sh.max() <= 1599 and abs(sh.mean()) < 80
    idx, off = ds._draw_background(100)
    assert (idx == 0).all() and off.min() >= 0 and off.max() < 80000


def test_transfer_learn_argument_contract():
    ms = input_data.standard_microspeech_model_settings(3)
    kw = dict(target="t", train_files=[], val_files=[], unknown_files=[], num_epochs=1, num_batches=1, batch_size=4,
              primary_lr=1e-3, embedding_lr=0, model_settings=ms, base_model_path="synthetic")
    with pytest.raises(ValueError):
        transfer_learning.transfer_learn(backprop_into_embedding=False, base_model_output="block7a_project_bn", **kw)   # 4-D cut: refused
    with pytest.raises(ValueError):
        transfer_learning.transfer_learn(backprop_into_embedding=True, base_model_output="dense_1", **kw)
    # backprop_into_embedding=True is implemented (tests/test_train_gpu.py); without a GPU it fails loudly like everything
    # else in the product path (no CPU fallback), never with NotImplementedError
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(Exception) as ei:
            transfer_learning.transfer_learn(backprop_into_embedding=True, base_model_output="dense_2", **kw)
        assert not isinstance(ei.value, NotImplementedError)
    assert transfer_learning.CATEGORIES == 3
    c, i = transfer_learning._split_confidences(np.array([[.1, .2, .7], [.6, .3, .1]]), 2).values()
    assert c == [0.7] and i == [0.6]


def test_synthetic_clip_generator_is_pinned():
    a = synth.clips_int16(3)
    assert a.shape == (3, 16000) and a.dtype == np.int16 and np.abs(a).max() <= 0.8 * 32767 + 1
    assert hashlib.sha1(a.astype("<i2").tobytes()).hexdigest()[:16] == hashlib.sha1(synth.clips_int16(3).astype("<i2").tobytes()).hexdigest()[:16]
    assert np.array_equal(synth.clips_int16(2, first_clip=1), a[1:3])
    f = synth.clips_float32(1)
    assert np.array_equal((f * 32768).astype(np.int16), a[:1])       # exactly what decode_wav would yield
    assert synth.clips_int16(4)[3, :5].tolist() == [-11190, -3043, -7679, 14245, 12554]


def test_stage_costs_match_survey_totals():
    c = arch.stage_costs(1)
    fused = lambda k: k.endswith("_front") or k.endswith("_block") or k in ("stem_dw", "stem_block1a")      # fused launches re-book their stages
    assert sum(f for k, (f, _) in c.items() if not fused(k)) - c["gap"][0] == 2 * 32974496
    assert c["block5b_block"][0] == sum(c["block5b" + sfx][0] for sfx in ("_expand", "_dw", "_gate", ""))
    assert c["block2a_front"][0] == c["block2a_expand"][0] + c["block2a_dw"][0]        # fused launch = both stages' flops
    assert c["block2a_front"][1] < c["block2a_expand"][1]                               # ...without the expanded tensor's bytes
    assert arch.FRONTEND_BYTES_PER_CLIP_F32 == 71840
    assert c["block2a_expand"][0] == 2 * 768000 and c["block7a"][0] == 2 * 1474560


def test_unknown_files_bank_and_sample_check(tmp_path):
    """run.py:259-278 on-disk conventions: unknown_files.txt listing, 1 s @ 16 kHz samples."""
    import util_data
    from multilingual_kws_amd.embedding import input_data
    bank = tmp_path / "unknown_words"
    rng = np.random.default_rng(0)
    for rel in ("en/clips/a/x.wav", "es/clips/b/y.wav"):
        util_data.write_wav(str(bank / rel), util_data.tone_clip(440, rng))
    (bank / "unknown_files.txt").write_text("en/clips/a/x.wav\nes/clips/b/y.wav\n")
    files = input_data.load_unknown_files(bank)
    assert files == [str(bank / "en/clips/a/x.wav"), str(bank / "es/clips/b/y.wav")]
    assert input_data.check_one_second_16k(files[0])
    util_data.write_wav(str(bank / "short.wav"), util_data.tone_clip(440, rng, n=8000))
    with pytest.raises(ValueError):
        input_data.check_one_second_16k(str(bank / "short.wav"))
    util_data.write_wav(str(bank / "r8k.wav"), util_data.tone_clip(440, rng, n=16000), rate=8000)
    with pytest.raises(ValueError):
        input_data.check_one_second_16k(str(bank / "r8k.wav"))
    with pytest.raises(FileNotFoundError):
        input_data.load_unknown_files(tmp_path)


def test_specaug_mask_table_for_any_number_of_masks():
    """Host draws for mkws_specaug_apply_n: [B, 2*(NF+NT)] table inside the image; the reference loops freq_n / time_n times for whatever
    SpecAugParams says (input_data.py:317-362), so more than two masks per axis are drawn, not refused or truncated."""
    from multilingual_kws_amd.embedding import input_data
    ms = input_data.standard_microspeech_model_settings(3)
    ds = input_data.AudioDataset(ms, ["t"], None, [], spec_aug_params=input_data.SpecAugParams(percentage=100), seed=0)
    m = ds._draw_specaug_masks(4096)
    assert m.shape == (4096, 8) and m.dtype == np.int32
    for k in range(2):
        fs, fz, ts, tz = m[:, 2 * k], m[:, 2 * k + 1], m[:, 4 + 2 * k], m[:, 5 + 2 * k]
        assert fz.min() == 0 and fz.max() == 2 and ((fs + fz) <= 40 - 1).all()        # start ~ U{0..40-size-1}
        assert tz.min() == 0 and tz.max() == 2 and ((ts + tz) <= 49 - 1).all()
    n_freq = (m[:, 1] > 0).astype(int) + (m[:, 3] > 0)
    assert abs((n_freq == 0).mean() - 1 / 3) < 0.05 and abs((n_freq == 2).mean() - 1 / 3) < 0.05   # freq_n ~ U{0,1,2}
    ds3 = input_data.AudioDataset(ms, ["t"], None, [], spec_aug_params=input_data.SpecAugParams(percentage=100, frequency_n_range=3, time_n_range=5,
                                                                                                  frequency_max_px=4, time_max_px=6), seed=1)
    m3 = ds3._draw_specaug_masks(6000)
    assert m3.shape == (6000, 2 * (3 + 5))
    nf = (m3[:, 1:6:2] > 0).sum(1)
    nt = (m3[:, 7::2] > 0).sum(1)
    assert set(nf) == {0, 1, 2, 3} and set(nt) == {0, 1, 2, 3, 4, 5}                    # freq_n ~ U{0..3}, time_n ~ U{0..5}
    assert all(abs((nf == k).mean() - 1 / 4) < 0.03 for k in range(4)) and all(abs((nt == k).mean() - 1 / 6) < 0.03 for k in range(6))
    assert m3[:, 1:6:2].max() == 4 and m3[:, 7::2].max() == 6
    assert ((m3[:, 0:6:2] + m3[:, 1:6:2]) <= 39).all() and ((m3[:, 6::2] + m3[:, 7::2]) <= 48).all()
    # masks are used in slot order: slot k is on only if freq_n > k
    assert ((m3[:, 3] > 0) <= (m3[:, 1] > 0)).all() and ((m3[:, 5] > 0) <= (m3[:, 3] > 0)).all()


def test_shipped_bn_calibration_file_is_what_the_recipe_produces():
    """multilingual_kws_amd/data/synthetic_bn_1234.npy (data shipped with the package) == tools/calibrate_synthetic_bn.py."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("calibrate_synthetic_bn", os.path.join(root, "tools", "calibrate_synthetic_bn.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    shipped = np.load(os.path.join(root, "multilingual_kws_amd", "data", "synthetic_bn_1234.npy"))
    assert shipped.dtype == np.float32 and np.array_equal(shipped, tool.calibrated_stats(1234))
    from multilingual_kws_amd import weights
    assert sum(t["count"] for t in weights.bn_stat_tensors()) == shipped.shape[0]
    with pytest.warns(RuntimeWarning, match="no calibrated BatchNorm statistics"):      # "synthetic:SEED" stays usable for any seed,
        other = weights.synthetic_blob(seed=99)                                          # but never SILENTLY uncalibrated
    assert np.array_equal(other, weights.synthetic_blob(seed=99, calibrate=False)) and other.shape == (weights.weight_count(),)


def test_single_clip_augment_follows_the_reference_law(tmp_path):
    """AudioDataset.augment(audio, label) and the _random_* helpers (reference input_data.py:277-304, 510-530): branch
    probabilities 10 % silence / 45 % unknown / 45 % target at transfer_learn's settings, silence = a background slice,
    unknown = one of the unknown files (shifted), target = shifted clip with background mixed in 80 % of the time."""
    d = make_fewshot_dataset(str(tmp_path), n_train=1, n_val=1, n_unknown=4, n_bg=2)
    ms = input_data.standard_microspeech_model_settings(3)
    ds = input_data.AudioDataset(ms, ["tiempo"], d["bg_dir"], d["unknown"], unknown_percentage=50.0, seed=3)
    clip, label = ds.get_single_target_waveforms(d["train"][0])
    assert label == "tiempo" and clip.shape == (16000,)
    unknown = np.stack([input_data._read_wav(f, 16000) for f in d["unknown"]])
    n, counts, mixed, shifts = 3000, {"_silence_": 0, "_unknown_": 0, "tiempo": 0}, 0, []
    for _ in range(n):
        a, lab = ds.augment(clip, label)
        assert a.shape == (16000,) and a.dtype == np.float32 and np.abs(a).max() <= 1.0
        counts[lab] += 1
        if lab == "tiempo":
            # the clean clip shifted by some k in [-1600, 1600) reproduces the output exactly unless background was mixed in
            is_mixed = True
            for k in range(-1600, 1600):
                sh = np.zeros(16000, np.float32)
                if k > 0:
                    sh[k:] = clip[:16000 - k]
                else:
                    sh[:16000 + k] = clip[-k:]
                if np.array_equal(sh, a):
                    is_mixed = False
                    shifts.append(k)
                    break
            mixed += is_mixed
        if counts["tiempo"] >= 60 and lab == "tiempo":
            break
    tot = sum(counts.values())
    assert abs(counts["_silence_"] / tot - 0.10) < 0.06 and abs(counts["_unknown_"] / tot - 0.45) < 0.12
    assert 0.6 < mixed / counts["tiempo"] < 0.95 and len(shifts) >= 3 and min(shifts) >= -1600 and max(shifts) <= 1599
    ds2 = input_data.AudioDataset(ms, ["tiempo"], d["bg_dir"], d["unknown"], unknown_percentage=50.0, time_shift_ms=0,
                                  background_frequency=0.0, silence_percentage=0.0, seed=4)
    labs = [ds2.augment(clip, label) for _ in range(200)]
    assert all(np.array_equal(a, clip) for a, l in labs if l == "tiempo")                 # nothing left to draw: identity
    assert all(any(np.array_equal(a, u) for u in unknown) for a, l in labs if l == "_unknown_")
    assert 0.35 < np.mean([l == "_unknown_" for _, l in labs]) < 0.65
    a, lab = ds._random_silence()
    assert lab == "_silence_" and a.shape == (16000,) and 0 < np.abs(a).max() <= 800 / 32768 * 6
    a, lab = ds._random_unknown()
    assert lab == "_unknown_" and any(np.array_equal(a, u) for u in unknown)
    extra = ds._random_silence_unknown(20)
    assert [l for _, l in extra] == ["_silence_"] * 2 + ["_unknown_"] * 10


def test_sqrt48_algorithm_matches_the_integer_definition():
    """The frontend's rounded square root for mel sums < 2^48 (mkws_frontend.hip sqrt48_round: float estimate + one exact integer correction)
    restated in numpy (tools/sqrt48_check.py) == bits.h Sqrt64 + round-to-nearest on random values and every boundary class, with the
    estimate forced off by up to +-4."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "sqrt48_check.py"), "20000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("mismatches 0") == 9


def test_call_tape_logs_and_passes_through():
    """embedding_trainer._CallTape (the recorder behind TrainStepGraph's default mode): every entry-point call goes through to the library
    object unchanged and is logged as (function, arguments) in call order, so that replaying the log re-issues the same calls."""
    from multilingual_kws_amd.embedding_trainer import _CallTape

    class FakeLib:
        def __init__(self):
            self.seen = []

        def mkws_op_a(self, x, y):
            self.seen.append(("a", x, y))
            return 0

        def mkws_op_b(self, z):
            self.seen.append(("b", z))
            return -3
    lib, log = FakeLib(), []
    tape = _CallTape(lib, log)
    assert tape.mkws_op_a(1, "p") == 0 and tape.mkws_op_b(7) == -3 and tape.mkws_op_a(2, "q") == 0
    assert lib.seen == [("a", 1, "p"), ("b", 7), ("a", 2, "q")]
    assert [a for _, a in log] == [(1, "p"), (7,), (2, "q")]
    lib.seen.clear()
    assert [f(*a) for f, a in log] == [0, -3, 0] and lib.seen == [("a", 1, "p"), ("b", 7), ("a", 2, "q")]
    with pytest.raises(AttributeError):
        tape.no_such_entry_point



def test_batch_groups_make_the_draws_of_the_step_by_step_stream(tmp_path):
    """input_data.BatchGroups.take(g) (g optimizer steps per forward pass of the frozen embedding, transfer_learning.FrozenHeadTrainer): the
    host draws -- shuffled indices, augmentation items, labels, SpecAugment masks -- are those of g consecutive single batches, in the same
    order, from the same generator state.  (Device assembly stubbed: what the kernels make of the tables is the -m gpu test's business.)"""
    d = make_fewshot_dataset(str(tmp_path), n_unknown=12)
    ms = input_data.standard_microspeech_model_settings(3)

    def stream(seed, bs):
        ds = input_data.AudioDataset(ms, ["target"], d["bg_dir"], d["unknown"], unknown_percentage=50.0,
                                     spec_aug_params=input_data.SpecAugParams(percentage=80), seed=seed)
        ds._assemble = lambda cds, drawn: drawn                     # host tables only
        ds.background_host = None                                    # (keeps get_background_data's sizes; nothing is uploaded)
        return ds.init_single_target(input_data.AUTOTUNE, d["train"], is_training=True).shuffle(1000).repeat().batch(bs)
    it = iter(stream(9, 7))
    single = [next(it)[0] for _ in range(9)]
    g = input_data.BatchGroups(stream(9, 7))
    grouped = g.take(4) + g.take(1) + g.take(4)
    assert len(grouped) == 9
    for (i1, l1, m1), (i2, l2, m2) in zip(single, grouped):
        assert i1.tobytes() == i2.tobytes() and np.array_equal(l1, l2) and np.array_equal(m1, m2)
    assert len({t[0].tobytes() for t in single}) == 9                # the batches differ from each other
    # the shuffled passes come from ONE Generator.permuted call per refill: the values and the generator state of consecutive
    # rng.permutation(n) calls (the stream of rounds 1-4: seeded runs keep their batches)
    for n, k in ((5, 103), (1, 4), (64, 3)):
        r1, r2 = np.random.default_rng(n), np.random.default_rng(n)
        assert np.array_equal(np.concatenate([r1.permutation(n) for _ in range(k)]), r2.permuted(np.tile(np.arange(n), (k, 1)), axis=1).reshape(-1))
        assert r1.integers(0, 1 << 30) == r2.integers(0, 1 << 30)
    ds = input_data.AudioDataset(ms, ["target"], d["bg_dir"], d["unknown"], seed=3)
    ref_rng = np.random.default_rng(3)
    g2 = input_data.BatchGroups(ds.init_single_target(input_data.AUTOTUNE, d["train"], is_training=True).shuffle(1000).repeat().batch(12))
    want = np.concatenate([ref_rng.permutation(len(d["train"])) for _ in range(3)])
    assert np.array_equal(g2._next_indices(), want[:12]) and np.array_equal(g2.buf, want[12:])
    with pytest.raises(ValueError):
        input_data.BatchGroups(input_data.ClipDataset(stream(1, 2).owner, d["train"], ["target"] * len(d["train"]), True).batch(2))   # no .repeat()
    assert transfer_learning.FORWARD_CLIPS == 3072 and transfer_learning.OVERLAP_FROM_GROUP == 4
    assert transfer_learning.steps_per_forward(512) == 6 and transfer_learning.steps_per_forward(64) == 48
    assert transfer_learning.steps_per_forward(2048) == 1 and transfer_learning.steps_per_forward(4096) == 1 and transfer_learning.steps_per_forward(512, 1024) == 2


def test_shared_embedding_key_is_the_base_weights_content_for_files_too(tmp_path, monkeypatch):
    """load_models_shared keys the shared handle on the CONTENT of the base weights: directories are walked, a single file (a
    Keras .h5 that model.json names by absolute path) is hashed itself -- two models on different .h5 files must not share an
    embedding -- and a missing base raises instead of hashing nothing."""
    import json
    loads = []

    class FakeEmb(str):
        device = "cpu"
    monkeypatch.setattr(transfer_learning, "load_base_model", lambda base, mb, cut: (loads.append(base) or (FakeEmb(f"emb{len(loads)}"), None)))
    monkeypatch.setattr(transfer_learning, "Head", lambda *a, **k: ("head", a[:3]))
    monkeypatch.setattr(transfer_learning, "TransferLearnedModel", lambda emb, head, blob, base: emb)
    bases = []
    for i, payload in enumerate([b"weights A", b"weights B", b"weights A"]):
        b = tmp_path / f"base{i}.h5"
        b.write_bytes(payload)
        bases.append(str(b))
    d = tmp_path / "dirbase"
    (d / "variables").mkdir(parents=True)
    (d / "variables" / "variables.index").write_bytes(b"weights A")
    paths = []
    for i, base in enumerate(bases + [str(d), str(tmp_path / "missing.h5")]):
        m = tmp_path / f"model{i}"
        m.mkdir()
        json.dump({"base_model_path": base, "base_model_output": "dense_2"}, open(m / "model.json", "w"))
        np.savez(m / "head.npz", dims=np.array([1024, 18, 3]), params=np.zeros(18507, np.float32))
        paths.append(str(m))
    models = transfer_learning.load_models_shared(paths[:4])
    assert models[0] == models[2] and models[0] != models[1]          # same bytes share, other bytes do not
    assert models[3] not in (models[0], models[1])                    # a directory with the same bytes inside is still another kind of base
    assert len(loads) == 3
    with pytest.raises(FileNotFoundError, match="does not exist"):
        transfer_learning.load_models_shared(paths[4:])
