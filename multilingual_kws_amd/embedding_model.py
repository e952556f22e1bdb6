"""Device embedding network handle: host wrapper over mkws_embed_* (include/mkws.h).

Stands in for the Keras sub-model `Model(base.inputs, base.get_layer("dense_2").output)` that
multilingual_kws/embedding/transfer_learning.py:36-43 builds: `.predict(x[B,49,40,1]) -> [B,1024]`.
"""
import ctypes

import numpy as np

from . import _lib

EMBEDDING_DIM = 1024
SPEC_SHAPE = (49, 40)
# Keras layer names a caller may cut the base model at (transfer_learning.py:38-42 uses get_layer(name=base_model_output)) ->
# (stage name of mkws_embed_forward_tap, features).  Flat outputs only: the reference puts Dense(18) straight on the cut.
OUTPUT_LAYERS = {"dense_2": ("dense_2", 1024), "dense_1": ("dense_1", 2048), "dense": ("dense", 2048),
                 "global_average_pooling2d": ("gap", 1280)}


class EmbeddingModel:
    def __init__(self, weight_blob, max_batch=1024, device=None, output="dense_2"):
        import torch
        if output not in OUTPUT_LAYERS:
            raise ValueError(f"base_model_output {output!r}: the embedding can be cut at {sorted(OUTPUT_LAYERS)} (flat outputs)")
        self.output, (self.output_stage, self.output_dim) = output, OUTPUT_LAYERS[output]
        self.L = _lib.lib()
        blob = np.ascontiguousarray(weight_blob, dtype=np.float32)
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.L.mkws_embed_create(blob.ctypes.data, blob.shape[0], int(max_batch), ctypes.byref(h)))
        self.h = h
        self.generation = _lib.next_generation()
        self.max_batch = int(max_batch)
        self._blob, self._replicas = blob, []

    def replicas(self, n):
        """[self, + n - 1 more handles on the same weights]: a handle owns one workspace, so batches that run CONCURRENTLY (the
        multi-lane serving graph of embedding.batch_streaming_analysis) need one handle each.  Created on first use, kept."""
        while len(self._replicas) < n - 1:
            self._replicas.append(EmbeddingModel(self._blob, self.max_batch, self.device, self.output))
        return [self] + self._replicas[:n - 1]

    def serving_lanes(self, n, plan_batch):
        """n handles for n batches in flight on n streams -- replicas only, never `self`: they run the PLAN of `plan_batch` clips (the clips the
        lanes hold together; option "plan_batch", include/mkws.h), which would slow down a lone eager call on the caller's own handle."""
        lanes = self.replicas(n + 1)[1:]
        for r in lanes:
            if r.get_option("plan_batch") != max(plan_batch, r.max_batch):
                r.set_option("plan_batch", max(plan_batch, r.max_batch))
        return lanes

    def close(self):
        for r in getattr(self, "_replicas", []):
            r.close()
        if getattr(self, "h", None):
            _lib.forget_graphs(self)
            self.L.mkws_embed_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _prep(self, spec):
        import torch
        if not torch.is_tensor(spec):
            spec = torch.as_tensor(np.asarray(spec, dtype=np.float32))
        spec = spec.to(self.device, dtype=torch.float32)
        if spec.dim() == 4 and spec.shape[-1] == 1:
            spec = spec[..., 0]
        if spec.dim() != 3 or tuple(spec.shape[1:]) != SPEC_SHAPE:
            raise ValueError(f"expected [B,49,40] or [B,49,40,1], got {tuple(spec.shape)}")
        return spec.contiguous()

    def forward(self, spec, out=None):
        """spec CUDA tensor [B,49,40(,1)] -> CUDA tensor [B,1024].  B may exceed max_batch (chunked)."""
        import torch
        spec = self._prep(spec)
        B = spec.shape[0]
        emb = out if out is not None else torch.empty((B, self.output_dim), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            for s in range(0, B, self.max_batch):
                n = min(self.max_batch, B - s)
                if self.output_stage == "dense_2":
                    self._call_forward(spec[s:s + n], n, emb[s:s + n])
                else:           # an earlier cut: the same launches, stopped at that layer
                    _lib.check(self.L.mkws_embed_forward_tap(self.h, ctypes.c_void_p(spec[s:s + n].data_ptr()), n, self.output_stage.encode(),
                                                             ctypes.c_void_p(emb[s:s + n].data_ptr()), n * self.output_dim, _lib.current_stream_ptr()))
        return emb

    def _call_forward(self, spec, n, emb):
        args = (self.h, ctypes.c_void_p(spec.data_ptr()), n, ctypes.c_void_p(emb.data_ptr()), _lib.current_stream_ptr())
        rc = self.L.mkws_embed_forward(*args)
        if rc == _lib.MKWS_ERR_EXCHANGE:
            # include/mkws.h, "Failure contract of the paired whole-block kernel": an EARLIER forward of this handle returned
            # NaN embeddings; the handle has switched plans and the repeated call is correct.  The earlier result cannot be
            # recalled from here, so say so loudly.
            import warnings
            warnings.warn("multilingual_kws_amd: " + self.L.mkws_last_error().decode("utf-8", "replace"), RuntimeWarning)
            rc = self.L.mkws_embed_forward(*args)
        _lib.check(rc)

    def checked(self, run):
        """run() -> host result of forward passes of this handle, synchronised (it ends in a device-to-host copy).  If one of them ran a
        failed pair / cluster exchange (include/mkws.h, failure contract: that forward's embeddings are NaN and the error is reported by the
        NEXT call) the handle is healed and run() repeated once, so that the numpy-facing API never hands back the poisoned batch.  The
        asynchronous device API (forward) keeps the documented contract: it cannot look without synchronising."""
        out = run()
        if self.get_option("exchange_error"):
            import warnings
            warnings.warn("multilingual_kws_amd: an in-kernel exchange failed in this forward pass; repeating it on the single-workgroup kernels", RuntimeWarning)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore", RuntimeWarning)      # (the repeat's first launch reports MKWS_ERR_EXCHANGE and is re-issued by _call_forward)
                out = run()
            if self.get_option("exchange_error"):
                raise _lib.MkwsError(_lib.MKWS_ERR_EXCHANGE, "an in-kernel exchange failed again on the healed handle")
        return out

    def predict(self, x):
        """Keras-style: numpy in ([B,49,40,1]), numpy out ([B,1024], or the width of the layer the model was cut at)."""
        return self.checked(lambda: self.forward(x).cpu().numpy())

    def tap(self, spec, stage):
        """Output of a named intermediate stage (flat float32 CUDA tensor) for parity debugging."""
        import torch
        spec = self._prep(spec)
        B = spec.shape[0]
        if B > self.max_batch:
            raise ValueError("tap: batch exceeds max_batch")
        dst = torch.empty(B * 48000, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            n = _lib.check(self.L.mkws_embed_forward_tap(self.h, ctypes.c_void_p(spec.data_ptr()), B, stage.encode(),
                                                         ctypes.c_void_p(dst.data_ptr()), dst.numel(), _lib.current_stream_ptr()))
        return dst[:n]

    def set_option(self, name, value):
        _lib.check(self.L.mkws_embed_set_option(self.h, name.encode(), int(value)))

    def get_option(self, name):
        return _lib.check(self.L.mkws_embed_get_option(self.h, name.encode()))

    def profile(self, spec, reps=5):
        """[(stage, kernel_name, avg_ms)] per kernel launch of one forward pass (hipEvent-timed)."""
        import torch
        spec = self._prep(spec)
        B = spec.shape[0]
        emb = torch.empty((B, EMBEDDING_DIM), dtype=torch.float32, device=self.device)
        buf = ctypes.create_string_buffer(1 << 16)
        with torch.cuda.device(self.device):
            _lib.check(self.L.mkws_embed_profile(self.h, ctypes.c_void_p(spec.data_ptr()), B, int(reps),
                                                 ctypes.c_void_p(emb.data_ptr()), buf, len(buf), _lib.current_stream_ptr()))
        rows = []
        for line in buf.value.decode().splitlines():
            stage, kernel, ms = line.split("\t")
            rows.append((stage, kernel, float(ms)))
        return rows
