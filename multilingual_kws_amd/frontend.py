"""Device micro-frontend handle: thin host wrapper over mkws_frontend_* (include/mkws.h)."""
import ctypes

import numpy as np

from . import _lib

_TABLE_IDS = dict(window_coef=(0, np.int16), twiddles=(1, np.int16), super_twiddles=(2, np.int16),
                  weights=(3, np.int16), unweights=(4, np.int16), chan_freq_starts=(5, np.int16),
                  chan_weight_starts=(6, np.int16), chan_widths=(7, np.int16), gain_lut=(8, np.int16),
                  log_lut=(9, np.uint16), scalars=(10, np.int32))
_SCALARS = ("window_size", "window_step", "fft_size", "start_index", "end_index", "num_weights",
            "snr_shift", "correction_bits")


def make_cfg(**over):
    cfg = _lib.FrontendCfg()
    _lib.lib().mkws_frontend_default_cfg(ctypes.byref(cfg))
    for k, v in over.items():
        if not hasattr(cfg, k):
            raise TypeError(f"unknown frontend option {k!r}")
        setattr(cfg, k, int(v) if isinstance(getattr(cfg, k), int) else float(v))
    return cfg


def host_table(cfg, name):
    """Host-only (works without a GPU): one of the integer tables the library builds for cfg."""
    which, dt = _TABLE_IDS[name]
    L = _lib.lib()
    n = _lib.check(L.mkws_frontend_host_table(ctypes.byref(cfg), which, None, 0))
    buf = np.zeros(n // np.dtype(dt).itemsize, dtype=dt)
    _lib.check(L.mkws_frontend_host_table(ctypes.byref(cfg), which, buf.ctypes.data, n))
    return buf


def host_scalars(cfg):
    return dict(zip(_SCALARS, host_table(cfg, "scalars").tolist()))


def num_frames(cfg, n_samples):
    return _lib.check(_lib.lib().mkws_frontend_num_frames(ctypes.byref(cfg), n_samples))


class Frontend:
    """One configured device frontend (tables uploaded once).  All tensors are torch CUDA tensors."""

    def __init__(self, max_samples=16000, **cfg_over):
        self.cfg = make_cfg(**cfg_over)
        self.L = _lib.lib()
        h = ctypes.c_void_p()
        _lib.check(self.L.mkws_frontend_create(ctypes.byref(self.cfg), int(max_samples), ctypes.byref(h)))
        self.h = h
        self.max_samples = int(max_samples)
        self.num_channels = self.cfg.num_channels

    def close(self):
        if getattr(self, "h", None):
            self.L.mkws_frontend_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def forward(self, audio, want_raw=False, out=None):
        """audio: CUDA tensor [B, n] float32 in [-1,1] or int16 PCM -> float32 [B, frames, channels]
        (= to_micro_spectrogram); with want_raw also the op's raw integers (int16 storage of uint16)."""
        import torch
        if not audio.is_cuda:
            raise ValueError("Frontend.forward needs a CUDA tensor (no CPU path)")
        if audio.dim() == 1:
            audio = audio[None]
        audio = audio.contiguous()
        B, n = audio.shape
        F = num_frames(self.cfg, n)
        spec = out if out is not None else torch.empty((B, F, self.num_channels), dtype=torch.float32, device=audio.device)
        raw = torch.empty((B, F, self.num_channels), dtype=torch.int16, device=audio.device) if want_raw else None
        rp = ctypes.c_void_p(raw.data_ptr()) if want_raw else None
        if B == 0 or F == 0:                  # empty in -> empty out, like the op
            return (spec, raw) if want_raw else spec
        if audio.dtype == torch.float32:
            fn = self.L.mkws_frontend_forward_f32
        elif audio.dtype == torch.int16:
            fn = self.L.mkws_frontend_forward_i16
        else:
            raise TypeError(f"audio must be float32 or int16, got {audio.dtype}")
        with torch.cuda.device(audio.device):
            _lib.check(fn(self.h, ctypes.c_void_p(audio.data_ptr()), B, n, ctypes.c_void_p(spec.data_ptr()), rp,
                          _lib.current_stream_ptr()))
        return (spec, raw) if want_raw else spec

    def stream(self, audio, window_samples, hop_samples, want_raw=False):
        """One long recording [n] float32 -> [num_windows, frames, channels], window w covering
        samples [w*hop, w*hop+window) -- batch_streaming_analysis.py:99-117 with frame sharing."""
        import torch
        audio = audio.contiguous()
        n = audio.shape[0]
        F = num_frames(self.cfg, window_samples)
        nw = 0 if n < window_samples else 1 + (n - window_samples) // hop_samples
        spec = torch.empty((nw, F, self.num_channels), dtype=torch.float32, device=audio.device)
        raw = torch.empty((nw, F, self.num_channels), dtype=torch.int16, device=audio.device) if want_raw else None
        if nw > 0:
            with torch.cuda.device(audio.device):
                got = _lib.check(self.L.mkws_frontend_stream_f32(
                    self.h, ctypes.c_void_p(audio.data_ptr()), n, window_samples, hop_samples,
                    ctypes.c_void_p(spec.data_ptr()), ctypes.c_void_p(raw.data_ptr()) if want_raw else None,
                    nw, _lib.current_stream_ptr()))
            assert got == nw
        return (spec, raw) if want_raw else spec
