"""Training-mode embedding network for `backprop_into_embedding=True`
(multilingual_kws/embedding/transfer_learning.py:94-112).

The reference sets `layer.trainable = True` on the nested embedding Model, which un-freezes EVERY layer inside it
(BatchNormalization included -- SURVEY.md section 3c), so the second `xfer.fit` runs keras/applications/efficientnet.py
with training=True: BatchNormalization normalises with batch statistics and updates its moving averages
(momentum 0.99, eps 1e-3), residual blocks apply drop-connect (Dropout with noise_shape (None,1,1,1), rate
0.2 * block_index / 16), and Adam(embedding_lr) updates every kernel, bias, gamma and beta.

Keras owns that graph in the reference; this module is its host-side counterpart: a forward tape and a backward
sweep over the C-ABI training operators of include/mkws.h (`mkws_op_*`, hand-written HIP in csrc/mkws_train.hip).
PyTorch only provides device memory, streams and torch.distributed.  Parameters, gradients and Adam moments are flat
device buffers in the weight blob's own order and Keras layouts, so the gradient all-reduce is a handful of
contiguous RCCL calls, issued as soon as a range of the buffer is final (the dense layers -- 93 % of the bytes --
finish first) so they overlap the rest of the backward sweep.

Round 3: every cross-workgroup reduction of the operators is fixed-order (partials in a scratch arena this class owns), so a step
is bit-reproducible; BatchNormalization's training forward is two launches instead of eight; the generic GEMM prefetches and
splits long reductions by itself; and `TrainStepGraph` captures forward + loss + backward + Adam of one step in a hipGraph (the
Adam step index lives on the device), so a step is one replay instead of ~1000 ctypes round trips.
"""
import ctypes
import os

import numpy as np

from . import _lib, weights
from .arch import BLOCKS

BN_EPS, BN_MOMENTUM, DROP_CONNECT_RATE = 1e-3, 0.99, 0.2
ACT_NONE, ACT_SWISH, ACT_RELU, ACT_SELU, ACT_SIGMOID = 0, 1, 2, 3, 4


def _down(h, w, k):
    pt, pl = k // 2 - (1 - h % 2), k // 2 - (1 - w % 2)
    return (h + pt + k // 2 - k) // 2 + 1, (w + pl + k // 2 - k) // 2 + 1, pt, pl


class EmbeddingTrainer:
    def __init__(self, weight_blob, device=None):
        import torch
        self.torch = torch
        self.L = _lib.lib()
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        blob = np.ascontiguousarray(weight_blob, dtype=np.float32)
        if blob.shape[0] != weights.weight_count():
            raise ValueError(f"blob has {blob.shape[0]} floats, architecture needs {weights.weight_count()}")
        self.tensors = {t["name"]: t for t in weights.manifest()}
        self.params = torch.from_numpy(blob.copy()).to(self.device)
        self.grads = torch.zeros_like(self.params)
        self.m = torch.zeros_like(self.params)
        self.v = torch.zeros_like(self.params)
        self.step_t = 0
        self.norm_mean = float(blob[self.tensors["normalization/mean"]["offset"]])
        self.norm_std = max(float(np.sqrt(blob[self.tensors["normalization/variance"]["offset"]])), 1e-7)
        self.tape = None
        self._pending = []
        # host-side caches: the tape allocates the same sequence of buffers every step, and parameter / gradient views are
        # fixed slices of the flat buffers -- re-creating either costs more host time than the small kernels take
        self._views = {}
        self._pool, self._pool_pos, self._pool_B = [], 0, None
        self._stream = None
        # partial sums of the fixed-order reductions (include/mkws.h: mkws_op_set_scratch); 16 Mi floats cover every layer
        self._scratch = torch.empty(16 << 20, dtype=torch.float32, device=self.device)
        # the trainer's own operator context (arena + deferred-fold queue): independent of other trainers on this thread and of the
        # thread it runs on; bound at the top of every public method (_bind_stream)
        self._ctx = ctypes.c_void_p()
        _lib.check(self.L.mkws_train_ctx_create(self._p(self._scratch), self._scratch.numel(), ctypes.byref(self._ctx)))
        self.d_step = torch.zeros(1, dtype=torch.int32, device=self.device)        # Adam step index for the graph-replayed step
        # Weight gradients off the critical path: nothing in the backward sweep reads a dW before the optimizer (or the all-reduce), and at
        # batch 64 the sweep is a chain of small launches that leaves most of the chip idle.  The weight-gradient launches (dW GEMMs,
        # depthwise / squeeze-excite / stem weight gradients) go to a second stream with its own operator context (scratch arena + fold
        # queue); the chain of input gradients stays on the caller's stream; the two join before the optimizer.  Capturable (fork / join).
        self.overlap_wgrad = os.environ.get("MKWS_TRAIN_WGRAD_STREAM", "1") != "0"
        self.wgrad_fork_every = max(1, int(os.environ.get("MKWS_TRAIN_WGRAD_FORK_EVERY", "1")))      # blocks of the sweep per fork
        # BatchNorm chunk statistics inside the producing launch: bit 0 = 1x1 convolutions (GEMM epilogue), bit 1 = 3x3 depthwise, bit 2 = 5x5
        # depthwise.  Same-call A/B (tools/gpu/r4_train6.sh, ms per step at batch 64 / 512): 0: 3.73-3.76 / 7.92, 1: 3.71-3.72 / 7.91,
        # 3: 3.72 / 7.90, 7: 3.76-3.77 / 7.98 -- 32 launches fewer per forward pass for 1 % of the step; the 5x5 depthwise loses (a thread of
        # the fused kernel owns eight rows x 25 taps)
        self.fuse_bn_stats = int(os.environ.get("MKWS_TRAIN_FUSE_BN_STATS", "3"))
        self._side = None
        self._scratch_side = None
        self._ctx_side = ctypes.c_void_p()

    def __del__(self):
        try:
            if getattr(self, "_ctx", None):
                self.L.mkws_train_ctx_destroy(self._ctx)
                self._ctx = None
            if getattr(self, "_ctx_side", None):
                self.L.mkws_train_ctx_destroy(self._ctx_side)
                self._ctx_side = None
        except Exception:
            pass

    # ---- plumbing -----------------------------------------------------------------------------------------------------
    def _s(self):
        return self._stream          # bound once per forward_train() / backward() / adam_step(): torch's current stream

    def _bind_stream(self):
        self._stream = _lib.current_stream_ptr()
        _lib.check(self.L.mkws_train_ctx_bind(self._ctx))                                          # a thread-local pointer store

    _STREAM = object()          # placeholder for "the stream this call is issued on" in a queued operator call

    def _wgrad(self, op, *args):
        """Queue one operator call -- weight-gradient launches only -- for the side stream.  args are evaluated now (pointers of this
        block's buffers), except _STREAM.  _wgrad_go issues the queue: one fork per block of the sweep (or per `wgrad_fork_every` blocks)
        instead of one per launch keeps the event traffic small."""
        if not self.overlap_wgrad:
            return _lib.check(op(*[self._s() if a is self._STREAM else a for a in args]))
        self._wgrad_q.append((op, args))

    def _wgrad_go(self, force=True):
        """Issue the queued weight-gradient launches on the side stream, ordered after everything queued on the caller's stream so far."""
        if not self._wgrad_q:
            return
        self._wgrad_skips += 1
        if not force and self._wgrad_skips < self.wgrad_fork_every:
            return
        self._wgrad_skips = 0
        q, self._wgrad_q = self._wgrad_q, []
        self._on_side(lambda: [_lib.check(op(*[self._s() if a is self._STREAM else a for a in args])) for op, args in q])

    def _on_side(self, fn):
        torch = self.torch
        if self._side is None:
            # the side stream only has to be done before the optimizer.  It runs at the DEFAULT priority, the same as the caller's stream: torch
            # offers no priority below the default (`priority_range()` is (0, -1) on ROCm too and positive values are clamped to 0), so nothing
            # is deprioritised here -- the two streams share the chip as the hardware queues arbitrate (profiles/r05_notes.md section 4).
            # MKWS_TRAIN_SIDE_PRIORITY = -1 raises the SIDE stream instead (an A/B knob; it lost).
            prio = os.environ.get("MKWS_TRAIN_SIDE_PRIORITY")
            self._side = torch.cuda.Stream(device=self.device, priority=min(int(prio), 0) if prio is not None else 0)
            self._scratch_side = torch.empty(16 << 20, dtype=torch.float32, device=self.device)
            _lib.check(self.L.mkws_train_ctx_create(self._p(self._scratch_side), self._scratch_side.numel(), ctypes.byref(self._ctx_side)))
            self._side_ptr = ctypes.c_void_p(self._side.cuda_stream)
        _lib.check(self.L.mkws_op_stream_wait(self._side_ptr, self._stream))       # (a library call, not torch's wait_stream: TrainStepGraph's tape replays it)
        _lib.check(self.L.mkws_train_ctx_bind(self._ctx_side))
        if not self._side_open:             # first use in this sweep: a fresh fold queue
            self._side_open = True
            _lib.check(self.L.mkws_op_fold_defer(1, self._side_ptr))
        main, self._stream = self._stream, self._side_ptr
        try:
            return fn()
        finally:
            self._stream = main
            _lib.check(self.L.mkws_train_ctx_bind(self._ctx))

    def _join_wgrad(self):
        """Everything the side stream holds becomes final and visible to the caller's stream."""
        self._wgrad_go()
        if not getattr(self, "_side_open", False):
            return
        self._side_open = False
        _lib.check(self.L.mkws_train_ctx_bind(self._ctx_side))
        _lib.check(self.L.mkws_op_fold_defer(0, self._side_ptr))
        _lib.check(self.L.mkws_train_ctx_bind(self._ctx))
        _lib.check(self.L.mkws_op_stream_wait(self._stream, self._side_ptr))

    _side_open = False
    _wgrad_q = ()
    _wgrad_skips = 0

    @staticmethod
    def _p(t):
        return ctypes.c_void_p(t.data_ptr()) if t is not None else None

    def P(self, name):
        v = self._views.get(("p", name))
        if v is None:
            t = self.tensors[name]
            v = self._views[("p", name)] = self.params[t["offset"]:t["offset"] + t["count"]]
        return v

    def G(self, name):
        v = self._views.get(("g", name))
        if v is None:
            t = self.tensors[name]
            v = self._views[("g", name)] = self.grads[t["offset"]:t["offset"] + t["count"]]
        return v

    def new(self, *shape):
        """Scratch / tape buffer.  Buffers are handed out from a pool in call order: one step's forward + backward makes the
        same sequence of requests as the previous one (same batch size), so after the first step nothing is allocated."""
        i = self._pool_pos
        self._pool_pos += 1
        if i < len(self._pool) and tuple(self._pool[i].shape) == tuple(shape):
            return self._pool[i]
        buf = self.torch.empty(shape, dtype=self.torch.float32, device=self.device)
        if i < len(self._pool):
            self._pool[i] = buf
        else:
            self._pool.append(buf)
        return buf

    def gemm(self, A, B, C, M, N, K, lda, ldb, ldc, ta=0, tb=0, acc=0, ksplit=0):
        # ksplit = 0: the library splits a long reduction over workgroups when the grid is small (fixed-order fold through the scratch)
        _lib.check(self.L.mkws_op_gemm(self._p(A), self._p(B), self._p(C), M, N, K, lda, ldb, ldc, ta, tb, acc, ksplit, self._s()))

    def blob(self):
        """Current parameters (incl. updated BatchNorm moving statistics) as a host weight blob."""
        return self.params.cpu().numpy()

    # ---- layers ---------------------------------------------------------------------------------------------------------
    def _bn_fwd(self, Z, M, C, prefix, act, res=None, row_scale=None, group=1):
        """Training-mode BN + activation; res / row_scale [M // group]: the residual branch A = row_scale * act(BN(Z)) + res in the same launch."""
        mean, var = self.new(C), self.new(C)
        A = self.new(M, C)
        # batch statistics + moving-average update (Keras does it during the training-mode forward pass) + normalise / activate
        _lib.check(self.L.mkws_op_bn_train_fwd_res(self._p(Z), M, C, self._p(self.P(prefix + "/gamma")), self._p(self.P(prefix + "/beta")), BN_EPS, act, BN_MOMENTUM,
                                                   self._p(self.P(prefix + "/moving_mean")), self._p(self.P(prefix + "/moving_variance")), self._p(mean), self._p(var),
                                                   self._p(A), self._p(res) if res is not None else None, self._p(row_scale) if row_scale is not None else None,
                                                   group, self._s()))
        return A, (Z, mean, var, M, C, prefix, act)

    def _bn_bwd(self, rec, dA, src="same", row_scale=None, bcast=None, bscale=0.0, group=1):
        """dLoss/dA -> dLoss/dZ in dA [M,C]; writes dgamma / dbeta.  The incoming gradient is src (default: dA itself; None: none)
        * row_scale[row // group] + bcast[row // group] * bscale, assembled inside the first launch."""
        Z, mean, var, M, C, prefix, act = rec
        if src == "same":
            src = dA
        _lib.check(self.L.mkws_op_bn_act_bwd_ex(self._p(Z), self._p(mean), self._p(var), self._p(self.P(prefix + "/gamma")), self._p(self.P(prefix + "/beta")), BN_EPS, act,
                                                self._p(dA), self._p(src) if src is not None else None, self._p(row_scale) if row_scale is not None else None,
                                                self._p(bcast) if bcast is not None else None, float(bscale), group,
                                                self._p(self.G(prefix + "/gamma")), self._p(self.G(prefix + "/beta")), M, C, self._s()))
        return dA

    def _conv_bn_fwd(self, X, M, K, N, wname, prefix, act, res=None, row_scale=None, group=1):
        """1x1 convolution + training-mode BN (+ activation / residual branch) as one operator: the GEMM's epilogue leaves the BN's chunk
        statistics whenever it can (mkws_op_conv_bn_fwd).  Returns (A, BN tape record) like _bn_fwd."""
        if not self.fuse_bn_stats & 1:
            return self._bn_fwd(self._conv_fwd(X, M, K, N, wname), M, N, prefix, act, res=res, row_scale=row_scale, group=group)
        Z, mean, var, A = self.new(M, N), self.new(N), self.new(N), self.new(M, N)
        _lib.check(self.L.mkws_op_conv_bn_fwd(self._p(X), self._p(self.P(wname)), self._p(Z), M, N, K, self._p(self.P(prefix + "/gamma")),
                                              self._p(self.P(prefix + "/beta")), BN_EPS, act, BN_MOMENTUM, self._p(self.P(prefix + "/moving_mean")),
                                              self._p(self.P(prefix + "/moving_variance")), self._p(mean), self._p(var), self._p(A),
                                              self._p(res) if res is not None else None, self._p(row_scale) if row_scale is not None else None, group, self._s()))
        return A, (Z, mean, var, M, N, prefix, act)

    def _conv_fwd(self, X, M, K, N, wname):
        Z = self.new(M, N)
        self.gemm(X, self.P(wname), Z, M, N, K, K, N, N)
        return Z

    def _conv_bwd(self, X, dZ, M, K, N, wname, need_dx=True, add_into=None):
        """dW = X^T dZ (the long reduction over the rows is split by the library), dX = dZ W^T (+ add_into, in place: the shortcut's
        gradient of a residual block joins in the GEMM epilogue instead of a separate launch)."""
        # every weight is used once per step: written, not accumulated (ksplit = 0: see gemm())
        self._wgrad(self.L.mkws_op_gemm, self._p(X), self._p(dZ), self._p(self.G(wname)), K, N, M, K, N, N, 1, 0, 0, 0, self._STREAM)
        if not need_dx:
            return None
        if add_into is not None:
            self.gemm(dZ, self.P(wname), add_into, M, K, N, N, N, K, ta=0, tb=1, acc=1)
            return add_into
        dX = self.new(M, K)
        self.gemm(dZ, self.P(wname), dX, M, K, N, N, N, K, ta=0, tb=1)
        return dX

    def _fc_fwd(self, X, M, K, N, prefix, act):
        Z, A = self.new(M, N), self.new(M, N)
        _lib.check(self.L.mkws_op_dense_fwd(self._p(X), self._p(self.P(prefix + "/kernel")), self._p(self.P(prefix + "/bias")), act, self._p(Z), self._p(A), M, N, K, self._s()))
        return A, (X, Z, M, K, N, prefix, act)

    def _fc_bwd(self, rec, dA, need_dx=True):
        X, Z, M, K, N, prefix, act = rec
        _lib.check(self.L.mkws_op_bias_act_bwd(self._p(Z), self._p(self.P(prefix + "/bias")), act, self._p(dA), self._p(self.G(prefix + "/bias")), M, N, self._s()))
        return self._conv_bwd(X, dA, M, K, N, prefix + "/kernel", need_dx)

    # ---- forward (training mode) ------------------------------------------------------------------------------------------
    def forward_train(self, spec, drop_masks=None, keep_scales=None):
        """spec CUDA [B,49,40(,1)] -> embedding CUDA [B,1024] in TRAINING mode; keeps the tape for backward().
        drop_masks: {block name: bool [B] of KEPT samples} for the residual blocks, or None (no drop-connect).
        keep_scales: the same information as device tensors {block name: float32 [B] = kept / (1 - rate)} that the caller owns and
        refills between steps (what a graph-replayed step needs: no host-to-device traffic inside the step)."""
        torch = self.torch
        self._bind_stream()
        spec = spec.to(self.device, dtype=torch.float32)
        if spec.dim() == 4:
            spec = spec[..., 0]
        spec = spec.contiguous()
        B = spec.shape[0]
        if self._pool_B != B:
            self._pool, self._pool_B = [], B
        self._pool_pos = 0
        tape = {"spec": spec, "B": B, "blocks": []}
        H, W = 25, 20
        Z = self.new(B * H * W, 32)
        _lib.check(self.L.mkws_op_stem_fwd(self._p(spec), self._p(self.P("stem_conv/kernel")), self.norm_mean, self.norm_std, self._p(Z), B, self._s()))
        x, tape["stem_bn"] = self._bn_fwd(Z, B * H * W, 32, "stem_bn", ACT_SWISH)
        for bi, (name, cin, cout, k, s, e) in enumerate(BLOCKS):
            p = "block" + name
            ce, se = cin * e, max(1, int(cin * 0.25))
            rec = {"name": name, "inp": x, "H": H, "W": W, "cin": cin, "cout": cout, "k": k, "s": s, "ce": ce, "se": se}
            Min = B * H * W
            if e != 1:
                Ae, rec["expand_bn"] = self._conv_bn_fwd(x, Min, cin, ce, p + "_expand_conv/kernel", p + "_expand_bn", ACT_SWISH)
            else:
                Ae = x
            if s == 2:
                Ho, Wo, pt, pl = _down(H, W, k)
            else:
                Ho, Wo, pt, pl = H, W, k // 2, k // 2
            rec.update(Ho=Ho, Wo=Wo, pt=pt, pl=pl, Ae=Ae)
            Mout = B * Ho * Wo
            # depthwise conv + BN + swish: the conv launch leaves the BN chunk statistics (mkws_op_dwconv_bn_fwd)
            if self.fuse_bn_stats & (2 if k == 3 else 4):
                Zd, dmean_, dvar_, Ad = self.new(Mout, ce), self.new(ce), self.new(ce), self.new(Mout, ce)
                _lib.check(self.L.mkws_op_dwconv_bn_fwd(self._p(Ae), self._p(self.P(p + "_dwconv/depthwise_kernel")), self._p(Zd), B, H, W, ce, k, s, pt, pl, Ho, Wo,
                                                        self._p(self.P(p + "_bn/gamma")), self._p(self.P(p + "_bn/beta")), BN_EPS, ACT_SWISH, BN_MOMENTUM,
                                                        self._p(self.P(p + "_bn/moving_mean")), self._p(self.P(p + "_bn/moving_variance")), self._p(dmean_),
                                                        self._p(dvar_), self._p(Ad), self._s()))
                rec["dw_bn"] = (Zd, dmean_, dvar_, Mout, ce, p + "_bn", ACT_SWISH)
            else:
                Zd = self.new(Mout, ce)
                _lib.check(self.L.mkws_op_dwconv_fwd(self._p(Ae), self._p(self.P(p + "_dwconv/depthwise_kernel")), self._p(Zd), B, H, W, ce, k, s, pt, pl, Ho, Wo,
                                                     self._s()))
                Ad, rec["dw_bn"] = self._bn_fwd(Zd, Mout, ce, p + "_bn", ACT_SWISH)
            # the squeeze-excite branch (pool, two 1x1 convolutions, excite multiply): two launches, a workgroup per (clip, channel slab)
            mean, Yr, R, Gt, As = self.new(B, ce), self.new(B, se), self.new(B, se), self.new(B, ce), self.new(Mout, ce)
            work = self.new(B, (ce + 127) // 128 * se)
            _lib.check(self.L.mkws_op_se_fwd(self._p(Ad), self._p(self.P(p + "_se_reduce/kernel")), self._p(self.P(p + "_se_reduce/bias")),
                                             self._p(self.P(p + "_se_expand/kernel")), self._p(self.P(p + "_se_expand/bias")),
                                             self._p(mean), self._p(Yr), self._p(R), self._p(Gt), self._p(As), self._p(work), B, Ho * Wo, ce, se, self._s()))
            rec.update(Ad=Ad, Gt=Gt, As=As, se_mean=mean, se_Yr=Yr, se_R=R, se_work=work)
            rec["residual"] = (s == 1 and cin == cout)
            if rec["residual"]:
                scale = self._views.get(("ones", B))
                if scale is None:
                    scale = self._views[("ones", B)] = torch.ones(B, dtype=torch.float32, device=self.device)
                if keep_scales is not None and name in keep_scales:
                    scale = keep_scales[name]
                elif drop_masks is not None and name in drop_masks:
                    rate = DROP_CONNECT_RATE * bi / len(BLOCKS)
                    scale = torch.as_tensor(np.asarray(drop_masks[name]), device=self.device).to(torch.float32) / (1.0 - rate)
                rec["keep_scale"] = scale.contiguous()
                # out = keep * BN(project) + shortcut, inside the BN launch
                x, rec["project_bn"] = self._conv_bn_fwd(As, Mout, ce, cout, p + "_project_conv/kernel", p + "_project_bn", ACT_NONE, res=x,
                                                         row_scale=rec["keep_scale"], group=Ho * Wo)
            else:
                x, rec["project_bn"] = self._conv_bn_fwd(As, Mout, ce, cout, p + "_project_conv/kernel", p + "_project_bn", ACT_NONE)
            tape["blocks"].append(rec)
            H, W = Ho, Wo
        HW = H * W
        tape["top_in"], tape["HW"] = x, HW
        At, tape["top_bn"] = self._conv_bn_fwd(x, B * HW, 320, 1280, "top_conv/kernel", "top_bn", ACT_SWISH)
        gap = self.new(B, 1280)
        _lib.check(self.L.mkws_op_pool_hw(self._p(At), self._p(gap), B, HW, 1280, self._s()))
        a, tape["dense"] = self._fc_fwd(gap, B, 1280, 2048, "dense", ACT_RELU)
        a, tape["dense_1"] = self._fc_fwd(a, B, 2048, 2048, "dense_1", ACT_RELU)
        emb, tape["dense_2"] = self._fc_fwd(a, B, 2048, 1024, "dense_2", ACT_SELU)
        self.tape = tape
        return emb

    # ---- backward ---------------------------------------------------------------------------------------------------------
    def backward(self, d_emb, allreduce=False):
        """d_emb CUDA [B,1024] = dLoss/dEmbedding of the forward_train() batch.  Fills self.grads (blob layout).
        allreduce: sum the gradient over torch.distributed ranks, overlapped with the sweep."""
        torch = self.torch
        tape = self.tape
        if tape is None:
            raise RuntimeError("backward() needs a forward_train() first")
        self._bind_stream()
        # second stages of the gradient reductions wait in a queue and run a batch at a time (mkws_op_fold_defer): nobody reads a weight
        # gradient before the all-reduce / the optimizer
        _lib.check(self.L.mkws_op_fold_defer(1, self._s()))
        done = False
        try:
            self._backward_sweep(tape, d_emb, allreduce)
            done = True
        finally:
            if not done:        # an operator raised mid-sweep: do not leave the contexts deferring (the queues are dropped by the next fold_defer(1))
                self.L.mkws_train_ctx_bind(self._ctx)
                self.L.mkws_op_fold_defer(0, self._s())
                try:
                    self._join_wgrad()
                except Exception:
                    self._side_open = False

    def _backward_sweep(self, tape, d_emb, allreduce):
        torch = self.torch
        B = tape["B"]
        # (no memset of the 52 MB gradient blob: every trainable tensor's gradient is WRITTEN by exactly one operator per step, and the slots
        #  of the non-trainable tensors -- moving statistics, normalisation constants -- are never touched after the zero-initialisation)
        self._pending = []
        self._wgrad_q = []
        # a working copy in a pooled buffer (the sweep overwrites it in place); a library launch, so that the recorded step replays it
        src = d_emb.to(self.device, dtype=torch.float32).contiguous()
        d = self.new(B, src.shape[1])
        ones = self._views.get(("ones", B))
        if ones is None:
            ones = self._views[("ones", B)] = torch.ones(B, dtype=torch.float32, device=self.device)
        _lib.check(self.L.mkws_op_row_scale_add(self._p(src), self._p(ones), None, self._p(d), B, src.shape[1], self._s()))
        d = self._fc_bwd(tape["dense_2"], d)
        self._wgrad_go()
        d = self._fc_bwd(tape["dense_1"], d)
        self._wgrad_go()
        if allreduce:      # dense_1 / dense_2 (25 M of the 52 MB) are final: their all-reduce runs under the rest of the sweep
            self._allreduce_range(self.tensors["dense_1/kernel"]["offset"], self.grads.shape[0])
        d = self._fc_bwd(tape["dense"], d)
        HW = tape["HW"]
        dAt = self.new(B * HW, 1280)
        dZt = self._bn_bwd(tape["top_bn"], dAt, src=None, bcast=d, bscale=1.0 / HW, group=HW)       # the pooled gradient spread over the pixels on the fly
        d = self._conv_bwd(tape["top_in"], dZt, B * HW, 320, 1280, "top_conv/kernel")
        self._wgrad_go()
        if allreduce:
            self._allreduce_range(self.tensors["top_conv/kernel"]["offset"], self.tensors["dense_1/kernel"]["offset"])
        for rec in reversed(tape["blocks"]):
            p = "block" + rec["name"]
            H, W, Ho, Wo, cin, cout, ce, se, k, s = (rec[q] for q in ("H", "W", "Ho", "Wo", "cin", "cout", "ce", "se", "k", "s"))
            Min, Mout = B * H * W, B * Ho * Wo
            d_out = d
            if rec["residual"]:
                # drop-connect scale applied while the BN backward reads the gradient; d_out stays intact for the shortcut
                dZp = self._bn_bwd(rec["project_bn"], self.new(Mout, cout), src=d_out, row_scale=rec["keep_scale"], group=Ho * Wo)
            else:
                dZp = self._bn_bwd(rec["project_bn"], d_out)
            dAs = self._conv_bwd(rec["As"], dZp, Mout, ce, cout, p + "_project_conv/kernel")
            # squeeze-excite backward: two launches for dAd / dmean / the pre-activation gradients + one for the four parameter gradients
            dAd, dmean, dYg, dYr = self.new(Mout, ce), self.new(B, ce), self.new(B, ce), self.new(B, se)
            _lib.check(self.L.mkws_op_se_bwd_fused(self._p(rec["Ad"]), self._p(rec["Gt"]), self._p(dAs), self._p(rec["se_mean"]), self._p(rec["se_Yr"]),
                                                   self._p(rec["se_R"]), self._p(self.P(p + "_se_reduce/kernel")), self._p(self.P(p + "_se_expand/kernel")),
                                                   self._p(dAd), self._p(dmean), self._p(dYg), self._p(dYr), None, None, None, None,
                                                   self._p(rec["se_work"]), B, Ho * Wo, ce, se, self._s()))
            self._wgrad(self.L.mkws_op_se_wgrad, self._p(rec["se_mean"]), self._p(rec["se_R"]), self._p(dYg), self._p(dYr),
                        self._p(self.G(p + "_se_reduce/kernel")), self._p(self.G(p + "_se_reduce/bias")),
                        self._p(self.G(p + "_se_expand/kernel")), self._p(self.G(p + "_se_expand/bias")), B, ce, se, self._STREAM)
            dZd = self._bn_bwd(rec["dw_bn"], dAd, bcast=dmean, bscale=1.0 / (Ho * Wo), group=Ho * Wo)     # + the squeeze's gradient, spread over the pixels
            dAe = self.new(Min, ce)
            self._wgrad(self.L.mkws_op_dwconv_bwd, self._p(rec["Ae"]), self._p(self.P(p + "_dwconv/depthwise_kernel")), self._p(dZd), None,
                        self._p(self.G(p + "_dwconv/depthwise_kernel")), B, H, W, ce, k, s, rec["pt"], rec["pl"], Ho, Wo, self._STREAM)
            _lib.check(self.L.mkws_op_dwconv_bwd(self._p(rec["Ae"]), self._p(self.P(p + "_dwconv/depthwise_kernel")), self._p(dZd), self._p(dAe),
                                                 None, B, H, W, ce, k, s, rec["pt"], rec["pl"], Ho, Wo, self._s()))
            if "expand_bn" in rec:
                dZe = self._bn_bwd(rec["expand_bn"], dAe)
                d_in = self._conv_bwd(rec["inp"], dZe, Min, cin, ce, p + "_expand_conv/kernel", add_into=d_out if rec["residual"] else None)
            else:
                d_in = dAe
                if rec["residual"]:
                    _lib.check(self.L.mkws_op_axpy(self._p(d_in), self._p(d_out), 1.0, d_in.numel(), self._s()))
            d = d_in
            self._wgrad_go(force=False)
        dZ0 = self._bn_bwd(tape["stem_bn"], d)
        self._wgrad(self.L.mkws_op_stem_bwd_weight, self._p(tape["spec"]), self._p(dZ0), self.norm_mean, self.norm_std, self._p(self.G("stem_conv/kernel")), B,
                    self._STREAM)
        _lib.check(self.L.mkws_op_fold_defer(0, self._s()))           # flush: every gradient is final from here on
        self._join_wgrad()
        if allreduce:
            self._allreduce_range(0, self.tensors["top_conv/kernel"]["offset"])
            for h in self._pending:
                h.wait()
            self._pending = []
        self.tape = None

    def _allreduce_range(self, lo, hi):
        import torch.distributed as dist
        self._wgrad_go()
        if getattr(self, "_side_open", False):                         # the range's gradients must be final before the collective reads them
            _lib.check(self.L.mkws_train_ctx_bind(self._ctx_side))
            _lib.check(self.L.mkws_op_fold_flush(self._side_ptr))
            _lib.check(self.L.mkws_train_ctx_bind(self._ctx))
            _lib.check(self.L.mkws_op_stream_wait(self._stream, self._side_ptr))
        _lib.check(self.L.mkws_op_fold_flush(self._s()))
        if dist.is_available() and dist.is_initialized():
            self._pending.append(dist.all_reduce(self.grads[lo:hi], op=dist.ReduceOp.SUM, async_op=True))

    def adam_step(self, lr, beta1=0.9, beta2=0.999, eps=1e-7, grad_scale=1.0):
        """Keras Adam over the whole blob (moving statistics / Normalization constants have zero gradient and stay put)."""
        self.step_t += 1
        self._bind_stream()
        _lib.check(self.L.mkws_op_adam(self._p(self.params), self._p(self.grads), self._p(self.m), self._p(self.v), self.params.shape[0], lr, beta1, beta2, eps,
                                       self.step_t, grad_scale, self._s()))

    def adam_step_dev(self, lr, beta1=0.9, beta2=0.999, eps=1e-7, grad_scale=1.0):
        """The same update with the step index read from self.d_step on the device (see TrainStepGraph)."""
        self._bind_stream()
        _lib.check(self.L.mkws_op_adam_dev(self._p(self.params), self._p(self.grads), self._p(self.m), self._p(self.v), self.params.shape[0], lr, beta1, beta2, eps,
                                           self._p(self.d_step), grad_scale, self._s()))

    def named_grads(self):
        g = self.grads.cpu().numpy()
        return {n: g[t["offset"]:t["offset"] + t["count"]].reshape(t["shape"]) for n, t in self.tensors.items()}


def drop_connect_rates():
    """{block name: drop rate} of the residual blocks (keras efficientnet: drop_connect_rate * block index / number of blocks)."""
    return {name: DROP_CONNECT_RATE * bi / len(BLOCKS) for bi, (name, cin, cout, k, st, e) in enumerate(BLOCKS) if st == 1 and cin == cout}


class _CallTape:
    """Stand-in for the ctypes library object that logs every entry-point call (function, converted arguments) while passing it through."""

    def __init__(self, lib, log):
        self._lib, self._log = lib, log

    def __getattr__(self, name):
        f = getattr(self._lib, name)
        log = self._log

        def call(*args):
            log.append((f, args))
            return f(*args)
        return call


class TrainStepGraph:
    """One optimizer step of the `backprop_into_embedding=True` phase -- training-mode forward, head loss / gradient, backward through
    the whole embedding, Keras Adam on the head and on the embedding -- recorded ONCE and replayed per step.

    Everything a step consumes sits in static device buffers the caller refills (spectrograms, labels, the per-block drop-connect
    scales); the Adam step index is a device counter (mkws_op_step_inc / *_adam*_dev), so the recorded launches are valid for every
    step.  Three ways to run the step (`mode`):
      "tape" (default)  the step's C-ABI calls (~560, with their converted arguments) are logged once and re-issued from a tight loop:
                        no tensor bookkeeping, no argument conversion -- and the weight gradients keep their own stream (round 4: a
                        hipGraph with the fork / join edges of that second stream replays SLOWER than the launch-by-launch step,
                        4.6-4.8 ms against 3.9-4.1 ms at batch 64; profiles/r04_notes.md);
      "hipgraph"        one hipGraph replay per step, single stream (use_graph=True);
      "eager"           the plain host-driven step (use_graph=False).
    All three issue the same launches with the same fixed-order sums: bit-identical parameters.
    Single process only: the data-parallel phase keeps the launch-by-launch path (its gradient all-reduce is host-driven)."""

    def __init__(self, trainer, head, batch, lr, use_graph=None, mode=None):
        import torch
        self.tr, self.head, self.B, self.lr = trainer, head, int(batch), float(lr)
        if mode is None:
            mode = "tape" if use_graph is None else ("hipgraph" if use_graph else "eager")
        if mode not in ("tape", "hipgraph", "eager"):
            raise ValueError(f"mode {mode!r}: 'tape', 'hipgraph' or 'eager'")
        self.mode = mode
        dev = trainer.device
        self.spec = torch.zeros((self.B, 49, 40), dtype=torch.float32, device=dev)
        self.labels = torch.zeros(self.B, dtype=torch.int32, device=dev)
        self.rates = drop_connect_rates()
        self.scales = {n: torch.ones(self.B, dtype=torch.float32, device=dev) for n in self.rates}
        self.graph, self.stats, self._tape, self._keep = None, None, None, []
        if mode == "eager":
            return
        # warm-up: fills the trainer's buffer pool for this batch size and every lazily created view, on state that is restored
        # afterwards (parameters, moving statistics, Adam moments, step counters)
        snap = [t.clone() for t in (trainer.params, trainer.m, trainer.v, trainer.d_step)]
        hstate, hstep = head.state_view().clone(), head.step_t          # parameters AND Adam moments of the head (set_params would zero m / v)

        def restore():
            torch.cuda.synchronize(dev)
            for t, c in zip((trainer.params, trainer.m, trainer.v, trainer.d_step), snap):
                t.copy_(c)
            head.state_view().copy_(hstate)
            head.step_t = hstep
        if mode == "hipgraph":
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                self._body(overlap=False)
            torch.cuda.current_stream(dev).wait_stream(side)
            restore()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.stats = self._body(overlap=False)
            self.graph = g
            return
        self._body()
        self._record()
        restore()

    def _record(self):
        """Log one step's library calls on the current stream (the pool buffers, views and streams they name stay put from here on)."""
        tr, head = self.tr, self.head
        log = []
        lib_t, lib_h = tr.L, head.L
        tr.L, head.L = _CallTape(lib_t, log), _CallTape(lib_h, log)
        try:
            self.stats = self._body()
        finally:
            tr.L, head.L = lib_t, lib_h
        self._tape = log
        self._tape_stream = _lib.current_stream_ptr().value
        self._tape_pool = tr._pool          # the buffers the tape names: held here, so they outlive a trainer that moves on to another batch size
        self._tape_options = self._options()

    def _options(self):
        """The trainer switches that decide WHICH launches a step is made of: a tape recorded under other values replays the old step."""
        tr = self.tr
        return (bool(tr.overlap_wgrad), int(tr.fuse_bn_stats), int(tr.wgrad_fork_every))

    def _body(self, overlap=None):
        tr, head = self.tr, self.head
        saved = tr.overlap_wgrad
        if overlap is not None:
            tr.overlap_wgrad = bool(overlap) and saved
        try:
            emb = tr.forward_train(self.spec, keep_scales=self.scales)
            stats = head.loss_grad(emb, self.labels)
            d_emb = head.input_grad(self.B)
            self._keep = [emb, d_emb]           # the recorded step names these buffers
            tr.backward(d_emb)
            _lib.check(tr.L.mkws_op_step_inc(tr._p(tr.d_step), _lib.current_stream_ptr()))
            head.adam_step_dev(self.lr, tr.d_step)
            tr.adam_step_dev(self.lr)
        finally:
            tr.overlap_wgrad = saved
        return stats

    def run(self, spec, labels, drop_masks=None):
        """spec CUDA [B,49,40(,1)], labels CUDA [B], drop_masks {block: bool [B] kept} (host) -> stats tensor [sum of row losses,
        #correct] of this step's batch (a view of the head's buffer: read it before the next step)."""
        import torch
        if spec.dim() == 4:
            spec = spec[..., 0]
        self.spec.copy_(spec)
        self.labels.copy_(labels.to(torch.int32))
        for n, rate in self.rates.items():
            if drop_masks is not None and n in drop_masks:
                keep = torch.as_tensor(np.asarray(drop_masks[n], dtype=np.float32) / np.float32(1.0 - rate))
                self.scales[n].copy_(keep, non_blocking=True)
            else:
                self.scales[n].fill_(1.0)
        if self.graph is not None:
            self.graph.replay()
            return self.stats
        if self._tape is not None:
            # raw ctypes calls: they launch on the CURRENT device's context, so make the trainer's device current as the eager wrappers do
            with torch.cuda.device(self.tr.device):
                if (_lib.current_stream_ptr().value != self._tape_stream or self.tr._pool is not self._tape_pool
                        or self._options() != self._tape_options):
                    # the caller moved to another stream, the trainer ran another batch size in between (its buffer pool was rebuilt), or one
                    # of its launch-plan switches changed: the tape names the old stream / buffers / launches.  Recording runs the step,
                    # so this one is done.
                    self._record()
                    return self.stats
                for f, a in self._tape:
                    rc = f(*a)
                    if rc is not None and rc < 0:
                        _lib.check(rc)
            return self.stats
        return self._body()
