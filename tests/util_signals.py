"""Deterministic test signals shared by oracle and GPU parity tests."""
import numpy as np


def lcg16k():
    x, out = 1, []
    for _ in range(16000):
        x = (1103515245 * x + 12345) & 0x7FFFFFFF
        out.append(((((x >> 8) & 0xFFFF) - 32768) * 8192) // 32768)
    return np.array(out, dtype=np.int16)


def d3_inputs():
    t = np.arange(16000)
    return {
        "square4": np.array([0, 32767, 0, -32768] * 4000, dtype=np.int16),
        "lcg": lcg16k(),
        "sine1k": np.round(16384 * np.sin(2 * np.pi * 1000 * t / 16000)).astype(np.int16),
        "zeros": np.zeros(16000, dtype=np.int16),
    }


def read_wav_pcm16(path, desired_samples=16000):
    """Minimal PCM16 mono reader for the golden clips (44-byte canonical header)."""
    w = open(path, "rb").read()
    n = int.from_bytes(w[40:44], "little") // 2
    pcm = np.zeros(desired_samples, dtype=np.int16)
    pcm[: min(n, desired_samples)] = np.frombuffer(w[44 : 44 + 2 * n], dtype="<i2")[:desired_samples]
    return pcm, w
