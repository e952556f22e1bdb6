"""ctypes binding of oracle/microfrontend_oracle.c (the CPU restatement of the TFLite-Micro
audio_microfrontend op behind multilingual_kws/embedding/input_data.py:19-35).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libmkws_oracle.so")


class MfoConfig(ctypes.Structure):
    _fields_ = [
        ("sample_rate", ctypes.c_int32),
        ("window_size_ms", ctypes.c_int32),
        ("window_step_ms", ctypes.c_int32),
        ("num_channels", ctypes.c_int32),
        ("upper_band_limit", ctypes.c_float),
        ("lower_band_limit", ctypes.c_float),
        ("smoothing_bits", ctypes.c_int32),
        ("even_smoothing", ctypes.c_float),
        ("odd_smoothing", ctypes.c_float),
        ("min_signal_remaining", ctypes.c_float),
        ("enable_pcan", ctypes.c_int32),
        ("pcan_strength", ctypes.c_float),
        ("pcan_offset", ctypes.c_float),
        ("gain_bits", ctypes.c_int32),
        ("enable_log", ctypes.c_int32),
        ("scale_shift", ctypes.c_int32),
    ]


def build(force=False):
    """Compile the C oracle with gcc (building the checker is not using it)."""
    src = os.path.join(_HERE, "microfrontend_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        L.mfo_create.restype = ctypes.c_void_p
        L.mfo_create.argtypes = [ctypes.POINTER(MfoConfig)]
        L.mfo_destroy.argtypes = [ctypes.c_void_p]
        L.mfo_num_frames.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.mfo_run_i16.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.mfo_run_batch_f32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        for name in ("window_size", "window_step", "fft_size", "start_index", "end_index", "num_weights",
                     "snr_shift", "correction_bits"):
            getattr(L, "mfo_" + name).argtypes = [ctypes.c_void_p]
            getattr(L, "mfo_" + name).restype = ctypes.c_int
        for name in ("window_coef", "weights", "unweights", "chan_freq_starts", "chan_weight_starts",
                     "chan_widths", "gain_lut", "log_lut", "twiddles", "super_twiddles"):
            getattr(L, "mfo_" + name).argtypes = [ctypes.c_void_p]
            getattr(L, "mfo_" + name).restype = ctypes.c_void_p
        L.mfo_window_frame.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.mfo_window_frame.restype = ctypes.c_int16
        L.mfo_noise_reduction.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.mfo_log.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
        L.mfo_log.restype = ctypes.c_uint32
        L.mfo_sqrt64.argtypes = [ctypes.c_uint64]
        L.mfo_sqrt64.restype = ctypes.c_uint32
        L.mfo_wide_dynamic_function.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
        L.mfo_wide_dynamic_function.restype = ctypes.c_int16
        L.mfo_fft.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        _lib = L
    return _lib


def _arr(ptr, n, dtype):
    buf = (ctypes.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype).copy()


class FrontendOracle:
    """One configured micro-frontend.  Defaults = what input_data.py:25-33 passes plus the Python
    wrapper's own defaults: 16 kHz, 30/20 ms, 40 channels, PCAN on."""

    def __init__(self, sample_rate=16000, window_size_ms=30, window_step_ms=20, num_channels=40,
                 upper_band_limit=7500.0, lower_band_limit=125.0, smoothing_bits=10,
                 even_smoothing=0.025, odd_smoothing=0.06, min_signal_remaining=0.05,
                 enable_pcan=True, pcan_strength=0.95, pcan_offset=80.0, gain_bits=21,
                 enable_log=True, scale_shift=6):
        self.cfg = MfoConfig(sample_rate, int(window_size_ms), int(window_step_ms), num_channels,
                             upper_band_limit, lower_band_limit, smoothing_bits, even_smoothing,
                             odd_smoothing, min_signal_remaining, int(bool(enable_pcan)), pcan_strength,
                             pcan_offset, gain_bits, int(bool(enable_log)), scale_shift)
        self.L = lib()
        self.h = self.L.mfo_create(ctypes.byref(self.cfg))
        if not self.h:
            raise ValueError("invalid micro-frontend configuration")
        self.num_channels = num_channels

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.L.mfo_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # -- tables --------------------------------------------------------------------------------
    def scalar(self, name):
        return getattr(self.L, "mfo_" + name)(self.h)

    def table(self, name):
        n1 = self.num_channels + 1
        sizes = {
            "window_coef": (self.scalar("window_size"), np.int16),
            "weights": (self.scalar("num_weights"), np.int16),
            "unweights": (self.scalar("num_weights"), np.int16),
            "chan_freq_starts": (n1, np.int16),
            "chan_weight_starts": (n1, np.int16),
            "chan_widths": (n1, np.int16),
            "gain_lut": (125, np.int16),
            "log_lut": (130, np.uint16),
            "twiddles": (self.scalar("fft_size"), np.int16),
            "super_twiddles": (self.scalar("fft_size") // 2, np.int16),
        }
        n, dt = sizes[name]
        return _arr(getattr(self.L, "mfo_" + name)(self.h), n, dt)

    # -- full pipeline -------------------------------------------------------------------------
    def num_frames(self, n):
        return self.L.mfo_num_frames(self.h, n)

    def run_i16(self, pcm, return_sig=False):
        """int16 [n] -> uint16 [frames, channels] (the op's raw output, out_scale=1)."""
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        F = self.num_frames(pcm.shape[0])
        out = np.zeros((F, self.num_channels), dtype=np.uint16)
        sig = np.zeros((F, self.num_channels), dtype=np.uint32) if return_sig else None
        self.L.mfo_run_i16(self.h, pcm.ctypes.data, pcm.shape[0], out.ctypes.data,
                           sig.ctypes.data if return_sig else None)
        return (out, sig) if return_sig else out

    def run_batch_f32(self, audio, nthreads=0, want_u16=False):
        """float32 [B, n] in [-1, 1] -> float32 [B, frames, channels] = to_micro_spectrogram."""
        audio = np.ascontiguousarray(audio, dtype=np.float32)
        if audio.ndim == 1:
            audio = audio[None]
        B, n = audio.shape
        F = self.num_frames(n)
        out = np.zeros((B, F, self.num_channels), dtype=np.float32)
        u16 = np.zeros((B, F, self.num_channels), dtype=np.uint16) if want_u16 else None
        self.L.mfo_run_batch_f32(self.h, audio.ctypes.data, B, n, out.ctypes.data,
                                 u16.ctypes.data if want_u16 else None, nthreads)
        return (out, u16) if want_u16 else out

    # -- per-stage taps ------------------------------------------------------------------------
    def window_frame(self, pcm):
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        out = np.zeros(self.scalar("window_size"), dtype=np.int16)
        m = self.L.mfo_window_frame(self.h, pcm.ctypes.data, out.ctypes.data)
        return out, int(m)

    def noise_reduction(self, estimate, signal):
        est = np.ascontiguousarray(estimate, dtype=np.uint32).copy()
        sig = np.ascontiguousarray(signal, dtype=np.uint32).copy()
        self.L.mfo_noise_reduction(self.h, est.ctypes.data, sig.ctypes.data)
        return est, sig

    def log(self, x):
        return int(self.L.mfo_log(self.h, x))

    def sqrt64(self, x):
        return int(self.L.mfo_sqrt64(x))

    def wide_dynamic_function(self, x):
        return int(self.L.mfo_wide_dynamic_function(self.h, x))

    def fft(self, frame):
        frame = np.ascontiguousarray(frame, dtype=np.int16)
        n = self.scalar("fft_size")
        assert frame.shape[0] == n
        out = np.zeros(n + 2, dtype=np.int16)
        self.L.mfo_fft(self.h, frame.ctypes.data, out.ctypes.data)
        return out.reshape(-1, 2)
