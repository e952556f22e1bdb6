"""Synthetic on-disk datasets for the pipeline tests (WAV files written with numpy only)."""
import os
import struct

import numpy as np


def wav_bytes(pcm16, rate=16000, channels=1):
    pcm16 = np.asarray(pcm16, dtype="<i2")
    data = pcm16.tobytes()
    hdr = b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE" + b"fmt " + struct.pack(
        "<IHHIIHH", 16, 1, channels, rate, rate * 2 * channels, 2 * channels, 16) + b"data" + struct.pack("<I", len(data))
    return hdr + data


def write_wav(path, pcm16, rate=16000, channels=1):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "wb") as f:
        f.write(wav_bytes(pcm16, rate, channels))


def tone_clip(freq, rng, n=16000, amp=9000, noise=1500, burst=None):
    t = np.arange(n)
    x = amp * np.sin(2 * np.pi * freq * t / 16000.0)
    if burst is not None:
        env = np.zeros(n)
        env[burst[0]:burst[1]] = 1.0
        x = x * env
    x = x + noise * rng.standard_normal(n)
    return np.clip(np.round(x), -32768, 32767).astype(np.int16)


def make_fewshot_dataset(root, n_train=5, n_val=8, n_unknown=24, n_bg=2, seed=0):
    """target = 2.2 kHz bursts, unknown = other tones; returns dict of file lists + bg dir."""
    rng = np.random.default_rng(seed)
    out = {"train": [], "val": [], "unknown": [], "bg_dir": os.path.join(root, "_background_noise_")}
    for i in range(n_train + n_val):
        p = os.path.join(root, "target", f"t{i}.wav")
        write_wav(p, tone_clip(2200 + 20 * rng.standard_normal(), rng, burst=(3000, 11000)))
        (out["train"] if i < n_train else out["val"]).append(p)
    for i in range(n_unknown):
        p = os.path.join(root, "other", f"u{i}.wav")
        write_wav(p, tone_clip(300 + 137 * (i % 9), rng, burst=(2000 + 100 * i, 9000 + 100 * i)))
        out["unknown"].append(p)
    for i in range(n_bg):
        write_wav(os.path.join(out["bg_dir"], f"bg{i}.wav"), (800 * rng.standard_normal(16000 * 6)).astype(np.int16))
    return out
