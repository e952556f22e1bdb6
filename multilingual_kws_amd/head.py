"""Device few-shot head handle: host wrapper over mkws_head_* (include/mkws.h).

Dense(18,tanh) -> Dense(3,softmax) + sparse CE + Keras Adam on the frozen embedding
(multilingual_kws/embedding/transfer_learning.py:47-59).
"""
import ctypes

import numpy as np

from . import _lib


def glorot_uniform_params(in_dim=1024, hidden=18, classes=3, seed=None):
    """Keras Dense defaults: kernel glorot_uniform, bias zeros.  Flat layout W1|b1|W2|b2."""
    rng = np.random.default_rng(seed)
    l1 = np.sqrt(6.0 / (in_dim + hidden))
    l2 = np.sqrt(6.0 / (hidden + classes))
    W1 = rng.uniform(-l1, l1, (in_dim, hidden))
    W2 = rng.uniform(-l2, l2, (hidden, classes))
    return np.concatenate([W1.ravel(), np.zeros(hidden), W2.ravel(), np.zeros(classes)]).astype(np.float32)


class Head:
    def __init__(self, in_dim=1024, hidden=18, classes=3, max_batch=1024, params=None, seed=None, device=None):
        import torch
        self.L = _lib.lib()
        self.in_dim, self.hidden, self.classes = in_dim, hidden, classes
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.L.mkws_head_create(in_dim, hidden, classes, int(max_batch), ctypes.byref(h)))
        self.h = h
        self.generation = _lib.next_generation()
        self.max_batch = int(max_batch)
        self.nparams = _lib.check(self.L.mkws_head_param_count(self.h))
        self.step_t = 0
        self._views = {}
        self._stats = torch.zeros(2, dtype=torch.float32, device=self.device)
        self.set_params(params if params is not None else glorot_uniform_params(in_dim, hidden, classes, seed))

    def close(self):
        if getattr(self, "h", None):
            _lib.forget_graphs(self)
            self.L.mkws_head_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_params(self, p):
        import torch
        p = np.ascontiguousarray(p, dtype=np.float32)
        with torch.cuda.device(self.device):
            _lib.check(self.L.mkws_head_set_params(self.h, p.ctypes.data, p.shape[0], _lib.current_stream_ptr()))
        self.step_t = 0

    def get_params(self):
        import torch
        p = np.zeros(self.nparams, dtype=np.float32)
        with torch.cuda.device(self.device):
            _lib.check(self.L.mkws_head_get_params(self.h, p.ctypes.data, self.nparams, _lib.current_stream_ptr()))
        return p

    def grad_view(self, with_stats=False):
        """The flat gradient buffer as a torch tensor aliasing the handle's device memory (for RCCL).
        with_stats: the all-reduce payload [P gradients | sum of row losses | #correct] (mkws_head_grad_count)."""
        n = _lib.check(self.L.mkws_head_grad_count(self.h)) if with_stats else self.nparams
        return self._alias(self.L.mkws_head_grads(self.h), n)

    def param_view(self):
        return self._alias(self.L.mkws_head_params(self.h), self.nparams)

    def state_view(self):
        """params | grads | Adam m | Adam v as ONE flat tensor aliasing the handle (snapshot / restore of the whole optimizer state)."""
        return self._alias(self.L.mkws_head_params(self.h), _lib.check(self.L.mkws_head_state_floats(self.h)))

    def _alias(self, ptr, n):
        import torch
        key = (int(ptr), int(n))
        t = self._views.get(key)
        if t is None:
            class _Holder:   # __cuda_array_interface__ producer over raw device memory
                pass
            hld = _Holder()
            hld.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f4", "data": (int(ptr), False), "version": 2}
            t = self._views[key] = torch.as_tensor(hld, device=self.device)
        return t

    def forward(self, emb):
        """emb CUDA [B,in] -> probs CUDA [B,classes]."""
        import torch
        emb = emb.contiguous()
        B = emb.shape[0]
        probs = torch.empty((B, self.classes), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.L.mkws_head_forward(self.h, ctypes.c_void_p(emb.data_ptr()), B, ctypes.c_void_p(probs.data_ptr()),
                                                _lib.current_stream_ptr()))
        return probs

    @staticmethod
    def forward_many(heads, emb):
        """N heads of equal dimensions over the same embeddings in one launch: emb CUDA [B,in] -> CUDA [N,B,classes]
        (multi-keyword serving on a shared embedding pass)."""
        import torch
        heads = list(heads)
        emb = emb.contiguous()
        B = emb.shape[0]
        h0 = heads[0]
        probs = torch.empty((len(heads), B, h0.classes), dtype=torch.float32, device=h0.device)
        table = (ctypes.c_void_p * len(heads))(*[h.h.value for h in heads])
        with torch.cuda.device(h0.device):
            _lib.check(h0.L.mkws_heads_forward(table, len(heads), ctypes.c_void_p(emb.data_ptr()), B,
                                               ctypes.c_void_p(probs.data_ptr()), _lib.current_stream_ptr()))
        return probs

    def loss_grad(self, emb, labels):
        """Fills the grad buffer with d(mean CE over these rows)/d(params); returns a CUDA tensor
        [2] = {sum of row losses, number correct} (asynchronous; .tolist() syncs)."""
        import torch
        emb = emb.contiguous()
        labels = labels.to(torch.int32).contiguous()
        with torch.cuda.device(self.device):
            _lib.check(self.L.mkws_head_loss_grad(self.h, ctypes.c_void_p(emb.data_ptr()), ctypes.c_void_p(labels.data_ptr()),
                                                  emb.shape[0], ctypes.c_void_p(self._stats.data_ptr()), _lib.current_stream_ptr()))
        return self._stats

    def input_grad(self, B):
        """After loss_grad on B rows: d(mean loss)/d(embedding rows), CUDA [B, in] (backprop_into_embedding)."""
        import torch
        dx = torch.empty((B, self.in_dim), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.L.mkws_head_input_grad(self.h, ctypes.c_void_p(dx.data_ptr()), B, _lib.current_stream_ptr()))
        return dx

    def reset_optimizer(self):
        """A fresh Adam (zero moments, t = 0) on the current parameters -- what re-compiling the Keras model does."""
        self.set_params(self.get_params())

    def adam_step_dev(self, lr, d_step, beta1=0.9, beta2=0.999, eps=1e-7, grad_scale=1.0):
        """Adam with the step index read from the int32 device tensor d_step (graph-replayed steps: embedding_trainer.TrainStepGraph)."""
        import torch
        with torch.cuda.device(self.device):
            _lib.check(self.L.mkws_head_adam_step_dev(self.h, lr, beta1, beta2, eps, ctypes.c_void_p(d_step.data_ptr()), grad_scale, _lib.current_stream_ptr()))

    def adam_step(self, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-7, grad_scale=1.0):
        import torch
        self.step_t += 1
        with torch.cuda.device(self.device):
            _lib.check(self.L.mkws_head_adam_step(self.h, lr, beta1, beta2, eps, self.step_t, grad_scale, _lib.current_stream_ptr()))
