"""Self-contained synthetic clips for benchmarks and parity tests.

clip b, sample t:  x = 0.45*sin(2*pi*f_b*t/16000) + 0.35*u,  f_b = 150 + 37*(b mod 128) Hz,
u in [-1,1) from splitmix64 (state seed + b, counter t; top 24 bits), quantised to int16
(round(x*32767)) and presented as float32 int16/32768 -- exactly what decode_wav would yield, so the
float->int16 cast inside to_micro_spectrogram never saturates.
"""
import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(state):
    """Vectorised splitmix64 output function on uint64 states (wraps mod 2^64)."""
    with np.errstate(over="ignore"):
        z = state.astype(np.uint64)
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def clips_int16(num_clips, num_samples=16000, seed=0x5EED0000, first_clip=0):
    b = (np.arange(num_clips, dtype=np.uint64) + np.uint64(first_clip))[:, None]
    t = np.arange(num_samples, dtype=np.uint64)[None, :]
    with np.errstate(over="ignore"):
        state = (np.uint64(seed) + b) * np.uint64(0x100000001B3) + (t + np.uint64(1)) * _GOLDEN
    u = (splitmix64(state) >> np.uint64(40)).astype(np.float64) / float(1 << 23) - 1.0
    f = 150.0 + 37.0 * (b % np.uint64(128)).astype(np.float64)
    x = 0.45 * np.sin(2.0 * np.pi * f * t.astype(np.float64) / 16000.0) + 0.35 * u
    return np.round(x * 32767.0).astype(np.int16)


def clips_float32(num_clips, num_samples=16000, seed=0x5EED0000, first_clip=0):
    return clips_int16(num_clips, num_samples, seed, first_clip).astype(np.float32) / np.float32(32768.0)


def wav_bytes(pcm16, rate=16000):
    """Mono PCM16 RIFF/WAVE file image (what decode_wav reads)."""
    import struct
    data = np.asarray(pcm16, dtype="<i2").tobytes()
    return (b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE" + b"fmt " +
            struct.pack("<IHHIIHH", 16, 1, 1, rate, rate * 2, 2, 16) + b"data" + struct.pack("<I", len(data)) + data)


def write_fewshot_dataset(root, n_target=5, n_val=8, n_unknown=256, n_bg=4, bg_seconds=60):
    """The synthetic fine-tune set of SURVEY.md section 8d (BASELINE configs[3]): n_target target clips,
    n_unknown "unknown word" clips and n_bg background tracks of bg_seconds, all from clips_int16 with distinct
    seeds, written as 16 kHz PCM16 WAV files in the reference's directory conventions.
    Returns {"train", "val", "unknown": file lists, "bg_dir"}."""
    import os
    out = {"train": [], "val": [], "unknown": [], "bg_dir": os.path.join(root, "_background_noise_")}

    def put(path, pcm):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "wb") as f:
            f.write(wav_bytes(pcm))
        return path
    tgt = clips_int16(n_target + n_val, seed=0x7A60E700)
    for i in range(n_target + n_val):
        (out["train"] if i < n_target else out["val"]).append(put(os.path.join(root, "target", f"t{i}.wav"), tgt[i]))
    unk = clips_int16(n_unknown, seed=0x0BAD0000)
    for i in range(n_unknown):
        out["unknown"].append(put(os.path.join(root, "unknown", f"u{i}.wav"), unk[i]))
    for i in range(n_bg):
        bg = clips_int16(1, num_samples=16000 * bg_seconds, seed=0xB6000000 + i)[0] // 8
        put(os.path.join(out["bg_dir"], f"bg{i}.wav"), bg)
    return out
