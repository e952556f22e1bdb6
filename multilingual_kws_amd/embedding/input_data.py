"""Drop-in for multilingual_kws/embedding/input_data.py on MI355X.

Same names, arguments and label conventions as the reference module; tensors are numpy / torch
(ROCm) instead of TensorFlow, and all sample-level work runs in libmkws_hip.so:

  to_micro_spectrogram  -> mkws_frontend_forward_*   (reference :19-35, the AudioMicrofrontend op)
  AudioDataset.augment  -> mkws_augment_batch        (reference :141-157, 227-304)
  spec_augment          -> mkws_specaug_apply_n      (reference :306-369)

Differences a caller can see: `AUTOTUNE` arguments are accepted and ignored; the dataset builders
return a small batch-producing object (`.shuffle().repeat().batch(n)`, iterable) instead of a
tf.data.Dataset, because batches are assembled on the GPU from waveform banks resident in HBM;
`to_micro_spectrogram` additionally accepts a batch `[B, samples]`.  Random augmentation follows the
reference's distributions (its tf.random.Generator stream itself is not reproducible elsewhere).
"""
import ctypes
import glob
import math
import os
import struct
from dataclasses import dataclass

import numpy as np

from .. import _lib
from ..frontend import Frontend

SILENCE_LABEL = "_silence_"
SILENCE_INDEX = 0
UNKNOWN_WORD_LABEL = "_unknown_"
UNKNOWN_WORD_INDEX = 1

AUTOTUNE = -1   # accepted wherever the reference passes tf.data.experimental.AUTOTUNE

# ---------------------------------------------------------------------------------------------------
# feature extraction
# ---------------------------------------------------------------------------------------------------
_frontends = {}


def _frontend_for(model_settings, max_samples):
    sr = model_settings["sample_rate"]
    wsize = int((model_settings["window_size_samples"] * 1000) / sr)
    wstep = int((model_settings["window_stride_samples"] * 1000) / sr)
    import torch
    key = (sr, wsize, wstep, model_settings["fingerprint_width"], torch.cuda.current_device())
    fe = _frontends.get(key)
    if fe is None or fe.max_samples < max_samples:
        fe = Frontend(max_samples=max(max_samples, 16000), sample_rate=sr, window_size_ms=wsize,
                      window_step_ms=wstep, num_channels=model_settings["fingerprint_width"])
        _frontends[key] = fe
    return fe


def to_micro_spectrogram(model_settings, audio):
    """float audio in [-1, 1], shape [samples] or [B, samples] -> [frames, channels] (or [B, ...]):
    the micro-frontend features scaled by 10/256.  numpy in -> numpy out; torch in -> CUDA torch out."""
    import torch
    is_np = not torch.is_tensor(audio)
    t = torch.as_tensor(np.asarray(audio, dtype=np.float32)) if is_np else audio
    if t.dtype not in (torch.float32, torch.int16):
        t = t.to(torch.float32)
    single = t.dim() == 1
    if single:
        t = t[None]
    if not t.is_cuda:
        t = t.cuda()
    out = _frontend_for(model_settings, t.shape[1]).forward(t)
    if single:
        out = out[0]
    return out.cpu().numpy() if is_np else out


def decode_wav(data, desired_samples=-1):
    """tf.audio.decode_wav(desired_channels=1, desired_samples=...) for 16-bit PCM: bytes ->
    (float32 [samples] = int16 / 32768, sample_rate); extra channels dropped; zero-padded or truncated."""
    if len(data) < 12 or data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError("not a RIFF/WAVE file")
    pos, fmt, pcm = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            fmt = struct.unpack("<HHIIHH", body[:16])
        elif cid == b"data":
            pcm = body
            break
        pos += 8 + size + (size & 1)
    if fmt is None or pcm is None:
        raise ValueError("WAV file lacks a fmt or data chunk")
    audio_format, channels, rate, _, _, bits = fmt
    if audio_format != 1 or bits != 16:
        raise ValueError(f"only 16-bit PCM WAV is supported (format {audio_format}, {bits} bits)")
    x = np.frombuffer(pcm[:len(pcm) // (2 * channels) * 2 * channels], dtype="<i2").reshape(-1, channels)[:, 0]
    x = x.astype(np.float32) / np.float32(32768.0)
    if desired_samples is not None and desired_samples > 0:
        if x.shape[0] >= desired_samples:
            x = x[:desired_samples]
        else:
            x = np.concatenate([x, np.zeros(desired_samples - x.shape[0], np.float32)])
    return x, rate


def _read_wav(path, desired_samples=-1):
    with open(path, "rb") as f:
        return decode_wav(f.read(), desired_samples)[0]


def load_unknown_files(unknown_words_dir, listing="unknown_files.txt"):
    """The unknown-words bank layout the reference's few-shot CLI consumes (run.py:272-278): a directory holding
    `unknown_files.txt`, one WAV path per line relative to that directory.  Returns absolute-ish path strings."""
    import os
    root = str(unknown_words_dir)
    lst = os.path.join(root, listing)
    if not os.path.isfile(lst):
        raise FileNotFoundError(f"{lst} not found")
    with open(lst, "r") as fh:
        return [os.path.join(root, w) for w in fh.read().splitlines() if w.strip()]


def check_one_second_16k(filepath):
    """What run.py:259-268 asks `soxi` (absent here) about every training sample: a 16 kHz, 1 s, PCM16 WAV.
    Raises ValueError otherwise."""
    with open(filepath, "rb") as f:
        audio, rate = decode_wav(f.read(), -1)
    if rate != 16000 or audio.shape[0] != 16000:
        raise ValueError(f"{filepath} appears to not be a 16KHz 1-second wav file ({rate} Hz, {audio.shape[0]} samples)")
    return True


def file2spec(model_settings, filepath):
    """WAV file -> spectrogram [frames, channels] (numpy); background-noise variant: AudioDataset.file2spec_w_bg."""
    audio = _read_wav(filepath, model_settings["desired_samples"])
    return to_micro_spectrogram(model_settings, audio)


def _next_power_of_two(x):
    return 1 if x == 0 else 2 ** (int(x) - 1).bit_length()


def prepare_model_settings(label_count, sample_rate, clip_duration_ms, window_size_ms, window_stride_ms,
                           feature_bin_count, preprocess):
    """Derived sizes shared by every consumer; key names are part of the interface."""
    desired_samples = int(sample_rate * clip_duration_ms / 1000)
    window_size_samples = int(sample_rate * window_size_ms / 1000)
    window_stride_samples = int(sample_rate * window_stride_ms / 1000)
    remaining = desired_samples - window_size_samples
    spectrogram_length = 0 if remaining < 0 else 1 + int(remaining / window_stride_samples)
    if preprocess == "average":
        fft_bin_count = 1 + (_next_power_of_two(window_size_samples) / 2)
        average_window_width = int(math.floor(fft_bin_count / feature_bin_count))
        fingerprint_width = int(math.ceil(fft_bin_count / average_window_width))
    elif preprocess in ("mfcc", "micro"):
        average_window_width = -1
        fingerprint_width = feature_bin_count
    else:
        raise ValueError('Unknown preprocess mode "%s" (should be "mfcc", "average", or "micro")' % (preprocess))
    return {
        "desired_samples": desired_samples,
        "window_size_samples": window_size_samples,
        "window_stride_samples": window_stride_samples,
        "spectrogram_length": spectrogram_length,
        "fingerprint_width": fingerprint_width,
        "fingerprint_size": fingerprint_width * spectrogram_length,
        "label_count": label_count,
        "sample_rate": sample_rate,
        "preprocess": preprocess,
        "average_window_width": average_window_width,
    }


def standard_microspeech_model_settings(label_count: int):
    return prepare_model_settings(label_count=label_count, sample_rate=16000, clip_duration_ms=1000,
                                  window_size_ms=30, window_stride_ms=20, feature_bin_count=40, preprocess="micro")


def add_background(foreground_audio, background_audio, background_volume):
    """Background scaled to the foreground's RMS, times background_volume, added and clipped to [-1, 1]."""
    import torch
    if torch.is_tensor(foreground_audio):
        fg, bg = foreground_audio, torch.as_tensor(background_audio, device=foreground_audio.device)
        fg_rms, bg_rms = torch.sqrt(torch.mean(fg * fg)), torch.sqrt(torch.mean(bg * bg))
        snr = torch.where(bg_rms > 0, fg_rms / bg_rms, torch.zeros_like(bg_rms))
        return torch.clamp((bg * snr) * background_volume + fg, -1.0, 1.0)
    fg = np.asarray(foreground_audio, dtype=np.float32)
    bg = np.asarray(background_audio, dtype=np.float32)
    fg_rms = np.sqrt(np.mean(np.square(fg), dtype=np.float32))
    bg_rms = np.sqrt(np.mean(np.square(bg), dtype=np.float32))
    snr = np.float32(fg_rms / bg_rms) if bg_rms > 0 else np.float32(0.0)
    return np.clip((bg * snr) * np.float32(background_volume) + fg, -1.0, 1.0).astype(np.float32)


@dataclass(frozen=True)
class SpecAugParams:
    percentage: float = 80.0
    frequency_n_range: int = 2     # number of frequency masks drawn from {0..n}
    frequency_max_px: int = 2      # each mask 1..max channels wide
    time_n_range: int = 2
    time_max_px: int = 2


class _AugItem(ctypes.Structure):   # mkws_augment_item
    _fields_ = [("mode", ctypes.c_int32), ("bank", ctypes.c_int32), ("src", ctypes.c_int32), ("shift", ctypes.c_int32),
                ("bg_idx", ctypes.c_int32), ("bg_off", ctypes.c_int32), ("bg_vol", ctypes.c_float), ("reserved", ctypes.c_int32)]


_ITEM_DTYPE = np.dtype([("mode", "<i4"), ("bank", "<i4"), ("src", "<i4"), ("shift", "<i4"),
                        ("bg_idx", "<i4"), ("bg_off", "<i4"), ("bg_vol", "<f4"), ("reserved", "<i4")])


class _HostStager:
    """One asynchronous host-to-device copy per batch for the small per-step tables (augmentation items, SpecAugment masks, labels).

    `torch.from_numpy(x).to(device)` copies from pageable memory: the call returns only when the copy has run, and the copy is queued
    behind everything already on the stream -- three of them per step made the fine-tune loop effectively synchronous (host 0.76 ms in
    next(it) against 0.66 ms of GPU work per 512-clip step; tools/finetune_host_profile.py).  Here the tables are packed into one pinned
    slot of a small ring, copied with non_blocking=True into a fresh device tensor and handed back as typed views; a slot is reused
    only after the copy that read it has executed (its event), so the host may run at most `slots` batches ahead."""

    def __init__(self, device, slots=8):
        self.device, self.slots, self.k = device, slots, 0
        self.host, self.events = [None] * slots, [None] * slots

    def upload(self, arrays):
        import torch
        offs, total = [], 0
        for a in arrays:
            offs.append(total)
            total += (a.nbytes + 15) & ~15
        i = self.k % self.slots
        self.k += 1
        if self.events[i] is not None:
            self.events[i].synchronize()
        if self.host[i] is None or self.host[i].numel() < total:
            self.host[i] = torch.empty(max(total, 1 << 16), dtype=torch.uint8).pin_memory()
        hv = self.host[i].numpy()
        for a, o in zip(arrays, offs):
            hv[o:o + a.nbytes] = np.ascontiguousarray(a).view(np.uint8).reshape(-1)
        dev = torch.empty(total, dtype=torch.uint8, device=self.device)
        dev.copy_(self.host[i][:total], non_blocking=True)
        ev = self.events[i] = self.events[i] or torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        return [dev[o:o + a.nbytes] for a, o in zip(arrays, offs)]


class ClipDataset:
    """What init_single_target / init_from_parent_dir / eval_with_silence_unknown return: a recipe for
    (spectrogram [B,frames,channels,1], label_id [B]) batches produced on the GPU."""

    def __init__(self, owner, files, labels, is_training, extra_silence=0, extra_unknown=0):
        self.owner, self.files, self.labels = owner, list(files), list(labels)
        self.label_ids = np.asarray([owner._label_id(l) for l in self.labels], dtype=np.int64)     # once, not per batch (512 Python calls a step)
        self.is_training = is_training
        self.extra_silence, self.extra_unknown = extra_silence, extra_unknown
        self._shuffle, self._repeat, self._batch = False, False, None
        self._bank = None

    def shuffle(self, buffer_size=None, **_):
        self._shuffle = True
        return self

    def repeat(self, count=None):
        self._repeat = True
        return self

    def batch(self, batch_size, **_):
        self._batch = int(batch_size)
        return self

    def prefetch(self, *_a, **_k):
        return self

    def __len__(self):
        return len(self.files) + self.extra_silence + self.extra_unknown

    def bank(self):
        if self._bank is None:
            self._bank = self.owner._upload_bank(self.files)
        return self._bank

    def __iter__(self):
        bs = self._batch or 1
        own, n = self.owner, len(self.files)
        if self._repeat:
            # endless stream (train_ds.shuffle().repeat().batch()): reshuffled passes cut into full batches
            groups = BatchGroups(self)
            while True:
                yield groups.take(1)
        order = own.rng.permutation(n) if self._shuffle else np.arange(n)
        todo = [("file", int(i)) for i in order] + [("sil", -1)] * self.extra_silence + [("unk", -1)] * self.extra_unknown
        for s in range(0, len(todo), bs):
            chunk = todo[s:s + bs]
            yield own._make_batch(self, np.asarray([i for k, i in chunk if k == "file"], dtype=np.int64),
                                  [k for k, _ in chunk if k != "file"])


class BatchGroups:
    """The endless training stream of a ClipDataset (`.shuffle().repeat().batch(bs)`), handed out G batches at a time.

    The few-shot fine-tune trains a head on a FROZEN embedding (transfer_learning.py:38-53 in the reference), so the forward pass of
    batch t + 1 does not depend on the optimizer step of batch t.  take(g) makes the host draws of the next g batches -- the same
    draws, in the same order, as g consecutive single batches -- and assembles them with ONE augmentation launch, ONE micro-frontend
    launch and ONE SpecAugment launch over g * bs clips (every one of those kernels is per-clip, so the clips are the ones the
    step-by-step stream produces, bit for bit).  The caller runs one embedding forward over the group and then g optimizer steps,
    each on its own bs rows."""

    def __init__(self, ds):
        if not ds._repeat:
            raise ValueError("BatchGroups needs an endless dataset: .repeat() before .batch()")
        self.ds, self.bs = ds, ds._batch or 1
        self.buf = np.zeros(0, dtype=np.int64)

    def _next_indices(self):
        own, n = self.ds.owner, len(self.ds.files)
        if n == 0:
            raise ValueError("cannot draw batches from an empty file list")
        if len(self.buf) < self.bs:         # reshuffled passes cut into full batches
            k = -(-(self.bs - len(self.buf)) // n)
            passes = np.tile(np.arange(n, dtype=np.int64), (k, 1))
            if self.ds._shuffle:
                # k passes in ONE generator call: permuted() shuffles the rows one after the other with the draws of k consecutive
                # rng.permutation(n) calls (same values, same generator state afterwards -- tests/test_host_logic.py); with 5 training
                # clips and 512-clip batches the 103 separate calls + concatenations of rounds 1-4 were 0.37 ms of host time per batch
                passes = own.rng.permuted(passes, axis=1)
            self.buf = np.concatenate([self.buf, passes.reshape(-1)])
        idx, self.buf = self.buf[:self.bs], self.buf[self.bs:]
        return idx

    def take(self, g=1):
        """-> (spectrograms [g * bs, frames, channels, 1], label ids [g * bs]); rows [j * bs, (j + 1) * bs) are batch j."""
        own = self.ds.owner
        return own._assemble(self.ds, [own._draw_batch(self.ds, self._next_indices(), []) for _ in range(int(g))])


class AudioDataset:
    def __init__(self, model_settings, commands, background_data_dir, unknown_files, time_shift_ms=100,
                 background_frequency=0.8, background_volume_range=0.1, silence_percentage=10.0,
                 unknown_percentage=10.0, spec_aug_params=SpecAugParams(), seed=None) -> None:
        self.model_settings = model_settings
        self._device = None
        self.rng = np.random.default_rng(seed)
        self.get_background_data(background_data_dir)
        self.max_time_shift_samples = self.timeshift_samples(time_shift_ms=time_shift_ms)
        self.background_frequency = background_frequency
        self.background_volume_range = background_volume_range
        # order-sensitive prepending so labels are always [silence, unknown, word1, word2, ...]
        commands = list(commands)
        self.unknown_percentage = unknown_percentage
        self.unknown_files = list(unknown_files)
        if len(self.unknown_files) > 0 and self.unknown_percentage > 0:
            commands = [UNKNOWN_WORD_LABEL] + commands
        self.silence_percentage = silence_percentage
        if self.silence_percentage > 0:
            commands = [SILENCE_LABEL] + commands
        self.commands = commands
        self.spec_aug_params = spec_aug_params
        self._unknown_bank = None
        self._unknown_loaded = {}

    # -- helpers ---------------------------------------------------------------------------------
    @property
    def device(self):
        """The GPU batches are assembled on (bound on first use; there is no CPU path)."""
        if self._device is None:
            import torch
            if not torch.cuda.is_available():
                raise RuntimeError("AudioDataset needs a GPU to produce batches (no CPU fallback)")
            self._device = torch.device("cuda", torch.cuda.current_device())
        return self._device

    def timeshift_samples(self, time_shift_ms=100):
        return int((time_shift_ms * self.model_settings["sample_rate"]) / 1000)

    def get_background_data(self, background_dir):
        """All *.wav under background_dir as one zero-padded [tracks, max_len] array + lengths
        (uploaded to the GPU on first use)."""
        tracks = []
        if background_dir is not None:
            for p in sorted(glob.glob(os.path.join(background_dir, "*.wav"))):
                tracks.append(_read_wav(p))
        self.background_sizes = np.asarray([t.shape[0] for t in tracks], dtype=np.int64)
        if tracks:
            bg = np.zeros((len(tracks), int(self.background_sizes.max())), dtype=np.float32)
            for i, t in enumerate(tracks):
                bg[i, :t.shape[0]] = t
            self.background_host = bg
        else:
            self.background_host = None
        self._background_dev = None

    @property
    def background_data(self):
        if self.background_host is None:
            return None
        if self._background_dev is None:
            import torch
            self._background_dev = torch.from_numpy(self.background_host).to(self.device)
        return self._background_dev

    def decode_audio(self, audio_binary):
        return decode_wav(audio_binary, self.model_settings["desired_samples"])[0]

    def get_label(self, file_path):
        return str(file_path).split(os.path.sep)[-2]

    def _label_id(self, label):
        # tf.argmax(label == commands): index of the first match, 0 when nothing matches
        return self.commands.index(label) if label in self.commands else 0

    def get_label_id_from_filename(self, filepath):
        return self._label_id(self.get_label(filepath))

    def _upload_bank(self, files):
        import torch
        n = self.model_settings["desired_samples"]
        arr = np.zeros((max(1, len(files)), n), dtype=np.float32)
        for i, f in enumerate(files):
            arr[i] = _read_wav(f, n)
        return torch.from_numpy(arr).to(self.device)

    def _unknown(self):
        if self._unknown_bank is None:
            self._unknown_bank = self._upload_bank(self.unknown_files)
        return self._unknown_bank

    # -- random draws (host, vectorised over the batch) ----------------------------------------------
    def _draw_shift(self, size=None):
        m = self.max_time_shift_samples
        if m <= 0:
            return 0 if size is None else np.zeros(size, dtype=np.int64)
        return int(self.rng.integers(-m, m)) if size is None else self.rng.integers(-m, m, size)

    def _draw_background(self, size=None):
        if self.background_sizes.shape[0] == 0:
            raise ValueError("augmentation needs background audio but background_data_dir held no *.wav")
        k = 1 if size is None else size
        idx = self.rng.integers(0, self.background_sizes.shape[0], k)
        hi = self.background_sizes[idx] - self.model_settings["desired_samples"]
        if (hi <= 0).any():
            raise ValueError("background track shorter than one clip")
        off = self.rng.integers(0, hi)
        return (int(idx[0]), int(off[0])) if size is None else (idx, off)

    def _draw_specaug_masks(self, B):
        """int32 [B, 2*(NF+NT)] mask table for mkws_specaug_apply_n (spec_augment / map_spec_aug distributions): NF = frequency_n_range
        channel masks then NT = time_n_range frame masks per clip, {start, size}, size 0 = unused.  The reference loops freq_n / time_n
        times for whatever SpecAugParams says (input_data.py:317-362); so does this table."""
        p = self.spec_aug_params
        NF, NT = int(p.frequency_n_range), int(p.time_n_range)
        frames, chans = self.model_settings["spectrogram_length"], self.model_settings["fingerprint_width"]
        masks = np.zeros((B, 2 * (NF + NT)), dtype=np.int32)
        apply = self.rng.uniform(0, 1, B) < p.percentage / 100
        freq_n = self.rng.integers(0, NF + 1, B)
        time_n = self.rng.integers(0, NT + 1, B)
        # draw order (kept from the two-slot table of rounds 1-3, so that seeded runs with the default parameters reproduce): slot k of the
        # frequency axis, then slot k of the time axis, for k = 0, 1, ...
        for k in range(max(NF, NT)):
            if k < NF:
                fsz = self.rng.integers(1, p.frequency_max_px + 1, B)
                fst = self.rng.integers(0, chans - fsz)
                on = apply & (freq_n > k)
                masks[:, 2 * k], masks[:, 2 * k + 1] = np.where(on, fst, 0), np.where(on, fsz, 0)
            if k < NT:
                tsz = self.rng.integers(1, p.time_max_px + 1, B)
                tst = self.rng.integers(0, frames - tsz)
                on = apply & (time_n > k)
                masks[:, 2 * (NF + k)], masks[:, 2 * (NF + k) + 1] = np.where(on, tst, 0), np.where(on, tsz, 0)
        return masks

    # -- batch assembly ------------------------------------------------------------------------------
    def _make_batch(self, ds, src_idx, extras):
        return self._assemble(ds, [self._draw_batch(ds, src_idx, extras)])

    def _draw_batch(self, ds, src_idx, extras):
        """Host side of one batch: EVERY random draw (augmentation items, labels, SpecAugment masks), nothing on the device.
        -> (items, labels, masks or None)."""
        nf, B = len(src_idx), len(src_idx) + len(extras)
        items = np.zeros(B, dtype=_ITEM_DTYPE)
        labels = np.zeros(B, dtype=np.int64)
        items["src"][:nf] = src_idx
        labels[:nf] = ds.label_ids[src_idx]
        if ds.is_training and nf > 0:
            # AudioDataset.augment: shift, then silence | unknown (shifted again) | background mix | as is
            have_unknown = len(self.unknown_files) > 0
            sil = self.rng.uniform(0, 1, nf) < self.silence_percentage / 100
            unk = ~sil & have_unknown & (self.rng.uniform(0, 1, nf) < self.unknown_percentage / 100)
            mix = ~sil & ~unk & (self.rng.uniform(0, 1, nf) < self.background_frequency)
            it = items[:nf]
            it["shift"] = self._draw_shift(nf)     # (the unknown branch re-shifts its replacement clip: same law)
            if have_unknown:
                it["src"] = np.where(unk, self.rng.integers(0, len(self.unknown_files), nf), it["src"])
            it["bank"] = unk
            it["mode"] = np.where(sil, 1, np.where(mix, 2, 0))
            if (sil | mix).any():
                bidx, boff = self._draw_background(nf)
                it["bg_idx"], it["bg_off"] = bidx, boff
                it["bg_vol"] = np.where(sil, self.rng.uniform(0, 1, nf), self.rng.uniform(0, self.background_volume_range, nf))
            labels[:nf] = np.where(sil, self._label_id(SILENCE_LABEL), np.where(unk, self._label_id(UNKNOWN_WORD_LABEL), labels[:nf]))
        for j, kind in enumerate(extras, start=nf):
            if kind == "sil":      # _random_silence
                items[j]["mode"], items[j]["bg_vol"] = 1, self.rng.uniform(0, 1)
                items[j]["bg_idx"], items[j]["bg_off"] = self._draw_background()
                labels[j] = self._label_id(SILENCE_LABEL)
            else:                  # _random_unknown
                items[j]["bank"], items[j]["src"] = 1, int(self.rng.integers(0, len(self.unknown_files)))
                labels[j] = self._label_id(UNKNOWN_WORD_LABEL)
        masks = self._draw_specaug_masks(B) if ds.is_training else None
        return items, labels, masks

    def _assemble(self, ds, drawn):
        """Device side of one batch -- or of several consecutive ones (BatchGroups): the drawn tables concatenated, ONE asynchronous
        copy (_HostStager), ONE launch each of augmentation, micro-frontend and SpecAugment over all their clips."""
        import torch
        n = self.model_settings["desired_samples"]
        items = drawn[0][0] if len(drawn) == 1 else np.concatenate([d[0] for d in drawn])
        labels = drawn[0][1] if len(drawn) == 1 else np.concatenate([d[1] for d in drawn])
        masks = None
        if any(d[2] is not None for d in drawn):
            width = max(d[2].shape[1] for d in drawn if d[2] is not None)
            masks = np.concatenate([d[2] if d[2] is not None else np.zeros((len(d[0]), width), np.int32) for d in drawn])     # size 0 = unused
        B = len(items)
        need_unknown = bool((items["bank"] == 1).any())
        L = _lib.lib()
        # host tables: every draw happened BEFORE the first launch, now ONE asynchronous copy (_HostStager)
        use_masks = masks is not None and bool(masks.any())
        if getattr(self, "_stager", None) is None:
            self._stager = _HostStager(self.device)
        with torch.cuda.device(self.device):
            up = self._stager.upload([items.view(np.uint8).reshape(-1), labels] + ([masks.reshape(-1)] if use_masks else []))
        d_items, d_labels = up[0], up[1].view(torch.int64)
        audio = torch.empty((B, n), dtype=torch.float32, device=self.device)
        bank0 = ds.bank()
        bank1 = self._unknown() if need_unknown else None
        bg = self.background_data
        with torch.cuda.device(self.device):
            _lib.check(L.mkws_augment_batch(
                ctypes.c_void_p(bank0.data_ptr()), ctypes.c_void_p(bank1.data_ptr()) if bank1 is not None else None,
                ctypes.c_void_p(bg.data_ptr()) if bg is not None else None, bg.shape[1] if bg is not None else 0,
                ctypes.c_void_p(d_items.data_ptr()), B, n, ctypes.c_void_p(audio.data_ptr()), _lib.current_stream_ptr()))
            spec = to_micro_spectrogram(self.model_settings, audio)
            self.last_masks = masks
            if use_masks:
                sp = self.spec_aug_params
                _lib.check(L.mkws_specaug_apply_n(ctypes.c_void_p(spec.data_ptr()), ctypes.c_void_p(up[2].data_ptr()), int(sp.frequency_n_range),
                                                  int(sp.time_n_range), B, spec.shape[1], spec.shape[2], _lib.current_stream_ptr()))
        self.last_audio = audio      # kept for tests / inspection
        return spec.unsqueeze(-1), d_labels

    # -- reference-shaped single-clip API (host-side, numpy) ----------------------------------------------
    def random_background_sample(self, background_volume=1.0):
        idx, off = self._draw_background()
        n = self.model_settings["desired_samples"]
        return (self.background_data[idx, off:off + n] * background_volume).reshape(n)

    def random_timeshift(self, audio):
        import torch
        a = self._draw_shift()
        t = torch.as_tensor(audio)
        n = self.model_settings["desired_samples"]
        out = torch.zeros(n, dtype=t.dtype, device=t.device)
        if a > 0:
            out[a:] = t[:n - a]
        else:
            out[:n + a] = t[-a:n]
        return out

    # -- single-clip augmentation exactly as the reference's tf.data map applies it (host-side numpy; the batch path above makes
    #    the same draws for a whole batch and runs the sample-level work in mkws_augment_batch) -----------------------
    def _background_sample_host(self, background_volume=1.0):
        idx, off = self._draw_background()
        n = self.model_settings["desired_samples"]
        return (self.background_host[idx, off:off + n] * np.float32(background_volume)).reshape(n).astype(np.float32)

    def _timeshift_host(self, audio):
        a, n = self._draw_shift(), self.model_settings["desired_samples"]
        audio = np.asarray(audio, dtype=np.float32)
        out = np.zeros(n, dtype=np.float32)
        if a > 0:
            out[a:] = audio[:n - a]
        else:
            out[:n + a] = audio[-a:n]
        return out

    def augment(self, audio, label):
        """(audio [desired_samples], label str) -> (audio, label): reference input_data.py:277-304 -- time shift, then with
        p = silence_percentage: a background slice at U(0,1) volume labelled _silence_; else with p = unknown_percentage (when
        unknown files exist): a random unknown-word clip, shifted again, labelled _unknown_; else with p =
        background_frequency: background mixed in at U(0, background_volume_range) of the clip's RMS."""
        audio = np.asarray(audio.cpu() if hasattr(audio, "cpu") else audio, dtype=np.float32)
        if self.max_time_shift_samples > 0:
            audio = self._timeshift_host(audio)
        if self.rng.uniform(0, 1) < self.silence_percentage / 100:
            background_volume = self.rng.uniform(0, 1)
            label = SILENCE_LABEL
            audio = self._background_sample_host(background_volume)
        elif len(self.unknown_files) > 0 and self.rng.uniform(0, 1) < self.unknown_percentage / 100:
            audio = self.get_unknown()
            if self.max_time_shift_samples > 0:
                audio = self._timeshift_host(audio)
            label = UNKNOWN_WORD_LABEL
        elif self.rng.uniform(0, 1) < self.background_frequency:
            background_volume = self.rng.uniform(0, self.background_volume_range)
            audio = add_background(audio, self._background_sample_host(), background_volume)
        return audio, label

    def _random_silence(self):
        """reference :510-514"""
        background_volume = self.rng.uniform(0, 1)
        return self._background_sample_host(background_volume), SILENCE_LABEL

    def _random_unknown(self):
        """reference :516-519"""
        return self.get_unknown(), UNKNOWN_WORD_LABEL

    def _random_silence_unknown(self, n_files):
        """reference :521-530: int(n * silence%) silence clips followed by int(n * unknown%) unknown-word clips."""
        n_silent = int(n_files * self.silence_percentage / 100)
        n_unknown = int(n_files * self.unknown_percentage / 100)
        return [self._random_silence() for _ in range(n_silent)] + [self._random_unknown() for _ in range(n_unknown)]

    def get_unknown(self):
        return _read_wav(self.unknown_files[int(self.rng.integers(0, len(self.unknown_files)))],
                         self.model_settings["desired_samples"])

    def get_waveform_and_label(self, file_path):
        return _read_wav(file_path, self.model_settings["desired_samples"]), self.get_label(file_path)

    def get_single_target_waveforms(self, file_path):
        return _read_wav(file_path, self.model_settings["desired_samples"]), self.commands[-1]

    def get_spectrogram_and_label_id(self, audio, label):
        return to_micro_spectrogram(self.model_settings, audio), self._label_id(label)

    def add_channel(self, spectrogram, label_id):
        return spectrogram[..., None], label_id

    def _add_bg(self, audio):
        import torch
        vol = float(self.rng.uniform(0, self.background_volume_range))
        return add_background(torch.as_tensor(audio, device=self.device), self.random_background_sample(), vol)

    def file2spec_w_bg(self, filepath):
        audio = _read_wav(filepath, self.model_settings["desired_samples"])
        return to_micro_spectrogram(self.model_settings, self._add_bg(audio)).cpu().numpy()

    def spec_augment(self, spectrogram):
        """Single spectrogram [frames, channels] (numpy or torch) -> masked copy (host-side convenience;
        batches go through mkws_specaug_apply)."""
        import torch
        p = self.spec_aug_params
        s = spectrogram.clone() if torch.is_tensor(spectrogram) else np.array(spectrogram, copy=True)
        frames, chans = s.shape[0], s.shape[1]
        for _ in range(int(self.rng.integers(0, p.frequency_n_range + 1))):
            size = int(self.rng.integers(1, p.frequency_max_px + 1))
            start = int(self.rng.integers(0, chans - size))
            s[:, start:start + size] = 0
        for _ in range(int(self.rng.integers(0, p.time_n_range + 1))):
            size = int(self.rng.integers(1, p.time_max_px + 1))
            start = int(self.rng.integers(0, frames - size))
            s[start:start + size, :] = 0
        return s

    def map_spec_aug(self, spectrogram, label_id):
        if self.rng.uniform(0, 1) < (self.spec_aug_params.percentage / 100):
            spectrogram = self.spec_augment(spectrogram)
        return spectrogram, label_id

    # -- dataset builders ------------------------------------------------------------------------------
    def init_single_target(self, AUTOTUNE, files, is_training):
        """Single-target model: every file is labelled with the target word (self.commands[-1])."""
        files = list(files)
        return ClipDataset(self, files, [self.commands[-1]] * len(files), is_training)

    def init_from_parent_dir(self, AUTOTUNE, files, is_training):
        """Label = name of the file's parent directory."""
        files = list(files)
        return ClipDataset(self, files, [self.get_label(f) for f in files], is_training)

    def eval_with_silence_unknown(self, AUTOTUNE, files, label_from_parent_dir: bool):
        files = list(files)
        if label_from_parent_dir:
            labels = [self.get_label(f) for f in files]
        else:
            assert len(self.commands) == 3, "model does not support both silence and unknown"
            labels = [self.commands[-1]] * len(files)
        n_silent = int(len(files) * self.silence_percentage / 100)
        n_unknown = int(len(files) * self.unknown_percentage / 100)
        return ClipDataset(self, files, labels, False, extra_silence=n_silent, extra_unknown=n_unknown)
