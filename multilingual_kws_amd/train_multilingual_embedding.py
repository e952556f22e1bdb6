"""Training the embedding model itself -- the reference's multilingual_kws/train_multilingual_embedding.py -- on MI355X.

In the reference this is a script with hard-coded paths: it builds (or re-loads) EfficientNetB0 + GAP + Dense 2048 / 2048 /
1024(selu) + Dense(num_labels) logits (:58-83), compiles it with Adam and SparseCategoricalCrossentropy(from_logits=True) (:84-93,
:99-104), and runs `model.fit(train_ds, validation_data=val_ds, epochs=EPOCHS, callbacks=[CSVLogger, ModelCheckpoint(filepath=
basename + ".{epoch:03d}-{val_accuracy:.4f}", monitor="val_accuracy", mode="max", save_best_only=True)])` (:106-131) on
`AudioDataset(model_settings, commands, bg_datadir, [], silence_percentage=1, unknown_percentage=0, SpecAugParams(percentage=80))
.init_from_parent_dir(...)` batches of 64 (:39-56), then pickles `history.history`.  Here the same loop is a function.

Everything numerical runs in the HIP operators of include/mkws.h: the embedding in TRAINING mode (batch-statistics BatchNorm with
moving-average updates, drop-connect: multilingual_kws_amd/embedding_trainer.py), the logits layer as `mkws_op_dense_fwd` +
`mkws_op_softmax_ce` + two `mkws_op_gemm` calls, Keras Adam over both; validation runs the inference kernels on the current weights.
Under torch.distributed (one process per GPU) every rank trains on its own shard of the files and the gradients are all-reduced
(the embedding's in three overlapped ranges, the classifier's in one more call).

This is the tail of SURVEY.md section 8 row f4; the dataset preparation around it (MSWC extraction, commands.txt / train_files.txt
bookkeeping) stays out of scope.
"""
import csv
import ctypes
import json
import os
import pickle

import numpy as np

from . import _lib, parallel, weights
from .embedding_model import EmbeddingModel
from .embedding_trainer import ACT_NONE, EmbeddingTrainer, drop_connect_rates


def checkpoint_name(basename, epoch, val_accuracy):
    """Keras ModelCheckpoint's formatting of `basename + ".{epoch:03d}-{val_accuracy:.4f}"` (epoch is 1-based there)."""
    return f"{basename}.{epoch:03d}-{val_accuracy:.4f}"


class LogitsLayer:
    """Dense(num_labels) on the 1024-D embedding (Keras defaults: glorot_uniform kernel, zero bias; no activation: the loss takes
    logits).  Parameters | gradients | Adam moments are flat device buffers  W[1024, N] | b[N]."""

    def __init__(self, num_labels, in_dim=1024, device=None, seed=None, params=None):
        import torch
        self.torch, self.L = torch, _lib.lib()
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.in_dim, self.n = int(in_dim), int(num_labels)
        if params is None:
            rng = np.random.default_rng(seed)
            lim = np.sqrt(6.0 / (self.in_dim + self.n))
            params = np.concatenate([rng.uniform(-lim, lim, self.in_dim * self.n), np.zeros(self.n)]).astype(np.float32)
        self.params = torch.from_numpy(np.ascontiguousarray(params, dtype=np.float32)).to(self.device)
        if self.params.numel() != self.in_dim * self.n + self.n:
            raise ValueError(f"logits layer needs {self.in_dim * self.n + self.n} parameters, got {self.params.numel()}")
        self.grads, self.m, self.v = (torch.zeros_like(self.params) for _ in range(3))
        self.W, self.b = self.params[:self.in_dim * self.n], self.params[self.in_dim * self.n:]
        self.dW, self.db = self.grads[:self.in_dim * self.n], self.grads[self.in_dim * self.n:]
        self._stats = torch.zeros(2, dtype=torch.float32, device=self.device)
        # the layer's own operator context (arena for the fixed-order reductions of the operators below; include/mkws.h, mkws_train_ctx)
        self._scratch = torch.empty(4 << 20, dtype=torch.float32, device=self.device)
        self._ctx = ctypes.c_void_p()
        _lib.check(self.L.mkws_train_ctx_create(self._p(self._scratch), self._scratch.numel(), ctypes.byref(self._ctx)))

    def __del__(self):
        try:
            if getattr(self, "_ctx", None):
                self.L.mkws_train_ctx_destroy(self._ctx)
                self._ctx = None
        except Exception:
            pass

    def _bind(self):
        _lib.check(self.L.mkws_train_ctx_bind(self._ctx))

    @staticmethod
    def _p(t):
        return ctypes.c_void_p(t.data_ptr()) if t is not None else None

    def forward(self, emb):
        """emb CUDA [B, in] -> logits CUDA [B, N]."""
        torch = self.torch
        emb = emb.contiguous()
        B = emb.shape[0]
        self._bind()
        Z, A = torch.empty((B, self.n), device=self.device), torch.empty((B, self.n), device=self.device)
        _lib.check(self.L.mkws_op_dense_fwd(self._p(emb), self._p(self.W), self._p(self.b), ACT_NONE, self._p(Z), self._p(A), B, self.n, self.in_dim,
                                            _lib.current_stream_ptr()))
        return A

    def loss_grad(self, emb, labels):
        """Mean sparse CE from logits over these rows: fills the gradient buffer, returns (stats [sum of row losses, #correct],
        d(mean loss)/d(emb) [B, in])."""
        torch = self.torch
        emb = emb.contiguous()
        B = emb.shape[0]
        s = _lib.current_stream_ptr()
        logits = self.forward(emb)
        rowstat = torch.empty((B, 2), device=self.device)
        labels = labels.to(torch.int32).contiguous()
        _lib.check(self.L.mkws_op_softmax_ce(self._p(logits), self._p(labels), B, self.n, self._p(rowstat), self._p(self._stats), s))
        d = logits                                                    # now d(mean loss)/d(logits)
        # db = column sums of d (bias_act_bwd with the identity activation leaves d untouched), dW = emb^T d, d_emb = d W^T
        _lib.check(self.L.mkws_op_bias_act_bwd(self._p(d), self._p(self.b), ACT_NONE, self._p(d), self._p(self.db), B, self.n, s))
        _lib.check(self.L.mkws_op_gemm(self._p(emb), self._p(d), self._p(self.dW), self.in_dim, self.n, B, self.in_dim, self.n, self.n, 1, 0, 0, 0, s))
        d_emb = torch.empty((B, self.in_dim), device=self.device)
        _lib.check(self.L.mkws_op_gemm(self._p(d), self._p(self.W), self._p(d_emb), B, self.in_dim, self.n, self.n, self.n, self.in_dim, 0, 1, 0, 0, s))
        return self._stats, d_emb

    def adam_step(self, lr, step_t, beta1=0.9, beta2=0.999, eps=1e-7, grad_scale=1.0):
        _lib.check(self.L.mkws_op_adam(self._p(self.params), self._p(self.grads), self._p(self.m), self._p(self.v), self.params.numel(), lr, beta1, beta2, eps,
                                       int(step_t), grad_scale, _lib.current_stream_ptr()))


class EmbeddingClassifier:
    """The Keras model of the reference's script: embedding network + logits layer.  predict() runs the inference kernels."""

    def __init__(self, blob, logits_params, num_labels, commands=None, device=None, max_batch=256):
        self.blob = np.ascontiguousarray(blob, dtype=np.float32)
        self.num_labels, self.commands = int(num_labels), list(commands) if commands is not None else None
        self.embedding = EmbeddingModel(self.blob, max_batch=max_batch, device=device)
        self.logits = LogitsLayer(num_labels, device=self.embedding.device, params=logits_params)

    def predict_device(self, spec):
        import torch
        outs = [self.logits.forward(self.embedding.forward(spec[s:s + self.embedding.max_batch])) for s in range(0, spec.shape[0], self.embedding.max_batch)]
        return torch.cat(outs) if outs else torch.empty((0, self.num_labels), device=self.embedding.device)

    def predict(self, x, batch_size=None, verbose=0):
        """numpy [N,49,40,1] (or [N,49,40]) -> numpy logits [N, num_labels]."""
        import torch
        x = torch.as_tensor(np.asarray(x, dtype=np.float32)).to(self.embedding.device)
        if x.dim() == 4:
            x = x[..., 0]
        return self.predict_device(x).cpu().numpy()

    def save(self, path):
        """A directory transfer_learn(base_model_path=...) accepts (the embedding as a weight container, as tf.keras' SavedModel of
        this model is what the reference's transfer_learn loads and cuts at dense_2) + the classifier layer."""
        weights.save(path, self.blob)
        np.savez(os.path.join(path, "logits.npz"), params=self.logits.params.cpu().numpy(), num_labels=self.num_labels)
        with open(os.path.join(path, "classifier.json"), "w") as f:
            json.dump({"format": "mkws-embedding-classifier-v1", "num_labels": self.num_labels, "commands": self.commands}, f)

    @classmethod
    def load(cls, path, device=None, max_batch=256):
        meta = json.load(open(os.path.join(path, "classifier.json")))
        z = np.load(os.path.join(path, "logits.npz"))
        return cls(weights.load(path), z["params"], int(meta["num_labels"]), meta.get("commands"), device=device, max_batch=max_batch)


def train_embedding(commands, train_files, val_files, bg_datadir, save_models_dir, epochs=8, batch_size=64, learning_rate=0.001,
                    base_checkpoint=None, basename="multilingual_context_", model_settings=None, steps_per_epoch=None, verbose=1, seed=None):
    """The reference's embedding training run as a function.  commands: the keyword list (commands.txt); train_files / val_files:
    wav paths whose parent directory is the label; base_checkpoint: a directory written by EmbeddingClassifier.save to resume from
    (the reference resumes `multilingual_context_.020-0.7058`), None = fresh weights with the initialisers of its model definition.
    steps_per_epoch: cap on the batches per epoch (None = one pass over train_files, as `fit` on a finite dataset).
    Returns (model, history) -- history has Keras' keys; checkpoints, CSV log and history pickle are written into save_models_dir
    with the reference's file names (rank 0 only)."""
    import torch
    from .embedding import input_data
    if not os.path.isdir(save_models_dir):
        raise ValueError("create model dir")                                    # reference :22-23
    if bg_datadir is None or not os.path.isdir(bg_datadir):
        raise ValueError("no bg data at", bg_datadir)                           # reference :36-37
    rank, world = parallel.rank(), parallel.world_size()
    ds_seed = None if seed is None else int(seed) + 1000003 * rank
    a = input_data.AudioDataset(model_settings or input_data.standard_microspeech_model_settings(label_count=len(commands) + 1), commands, bg_datadir, [],
                                silence_percentage=1, unknown_percentage=0, spec_aug_params=input_data.SpecAugParams(percentage=80), seed=ds_seed)
    num_labels = len(a.commands)                                                # includes _silence_
    if model_settings is not None:
        assert num_labels == model_settings["label_count"]                      # reference :62
    AUTOTUNE = input_data.AUTOTUNE
    # Data parallel: every rank must take the same number of steps with the same batch sizes (each step ends in collectives, and the
    # gradient average weights the ranks equally), so the shards are EQUAL: the len(train_files) % world leftover files are dropped.
    all_train = list(train_files)
    per_rank = len(all_train) // world
    if per_rank == 0:
        raise ValueError(f"{len(all_train)} training files cannot be sharded over {world} ranks")
    train_ds = a.init_from_parent_dir(AUTOTUNE, all_train[rank::world][:per_rank], is_training=True).shuffle(buffer_size=8000).batch(batch_size)
    val_ds = a.init_from_parent_dir(AUTOTUNE, val_files, is_training=False).batch(batch_size)

    if base_checkpoint is not None:
        prev = EmbeddingClassifier.load(base_checkpoint)
        blob, logit_params = prev.blob, prev.logits.params.cpu().numpy()
        if prev.num_labels != num_labels:
            raise ValueError(f"checkpoint has {prev.num_labels} labels, this run {num_labels}")
        del prev
    else:
        blob = _fresh_blob(seed)
        logit_params = None
    trainer = EmbeddingTrainer(blob)
    logits = LogitsLayer(num_labels, device=trainer.device, seed=seed, params=logit_params)
    if world > 1:                                                               # identical start on every rank
        parallel.broadcast_(trainer.params, 0)
        parallel.broadcast_(logits.params, 0)
    rates = drop_connect_rates()
    history = {"loss": [], "accuracy": [], "val_loss": [], "val_accuracy": []}
    log_idx = 0
    while os.path.isfile(os.path.join(save_models_dir, f"{basename}_log_{log_idx}.csv")):
        log_idx += 1
    csvlog_dest = os.path.join(save_models_dir, f"{basename}_log_{log_idx}.csv")
    best, step_t, model = -np.inf, 0, None
    for epoch in range(epochs):
        acc_stats = torch.zeros(2, dtype=torch.float64, device=trainer.device)
        seen = 0
        for bi, (spec, labels) in enumerate(train_ds):
            if steps_per_epoch is not None and bi >= steps_per_epoch:
                break
            nb = spec.shape[0]
            masks = {name: a.rng.uniform(0, 1, nb) >= rate for name, rate in rates.items()}
            emb = trainer.forward_train(spec, masks)
            stats, d_emb = logits.loss_grad(emb, labels)
            trainer.backward(d_emb, allreduce=world > 1)
            if world > 1:
                parallel.allreduce_sum_(logits.grads)
                stats = parallel.allreduce_sum_(stats.clone())
            step_t += 1
            trainer.adam_step(lr=learning_rate, grad_scale=1.0 / world)
            logits.adam_step(learning_rate, step_t, grad_scale=1.0 / world)
            acc_stats += stats.to(torch.float64)
            seen += nb * world
        tl, ta = (acc_stats / max(seen, 1)).tolist()
        # validation on the current weights (moving statistics), inference kernels; every rank evaluates the whole validation set
        model = EmbeddingClassifier(trainer.blob(), logits.params.cpu().numpy(), num_labels, a.commands, device=trainer.device, max_batch=max(batch_size, 64))
        vstats, vseen = np.zeros(2), 0
        for spec, labels in val_ds:
            z = model.predict_device(spec[..., 0])
            lab = labels.long()
            vstats[0] += float(torch.nn.functional.cross_entropy(z, lab, reduction="sum"))
            vstats[1] += float((z.argmax(1) == lab).sum())
            vseen += len(lab)
        vl, va = (vstats / max(vseen, 1)).tolist()
        for k, v in zip(("loss", "accuracy", "val_loss", "val_accuracy"), (tl, ta, vl, va)):
            history[k].append(v)
        if verbose and rank == 0:
            print(f"Epoch {epoch + 1}/{epochs} - loss: {tl:.4f} - accuracy: {ta:.4f} - val_loss: {vl:.4f} - val_accuracy: {va:.4f}")
        if rank == 0:
            with open(csvlog_dest, "w", newline="") as f:                        # CSVLogger(append=False): rewritten with every finished epoch
                w = csv.writer(f)
                w.writerow(["epoch", "accuracy", "loss", "val_accuracy", "val_loss"])
                for e in range(epoch + 1):
                    w.writerow([e, history["accuracy"][e], history["loss"][e], history["val_accuracy"][e], history["val_loss"][e]])
            if va > best:                                                       # ModelCheckpoint(monitor="val_accuracy", mode="max", save_best_only=True)
                best = va
                model.save(os.path.join(save_models_dir, checkpoint_name(basename, epoch + 1, va)))
    if rank == 0:
        history_idx = 0
        while os.path.isfile(os.path.join(save_models_dir, f"history_keras_{history_idx}.pkl")):
            history_idx += 1
        with open(os.path.join(save_models_dir, f"history_keras_{history_idx}.pkl"), "wb") as fh:
            pickle.dump(history, fh)
    return model, history


def _fresh_blob(seed):
    """Fresh weights with the initialisers of the reference's model definition (weights.synthetic_blob: VarianceScaling conv kernels,
    glorot_uniform / lecun_normal dense kernels) and Keras' fresh BatchNorm state: gamma 1, beta 0, moving mean 0, moving variance 1,
    zero biases."""
    blob = weights.synthetic_blob(weights.DEFAULT_SEED if seed is None else int(seed), calibrate=False)
    for t in weights.manifest():
        leaf = t["name"].split("/")[-1]
        sl = slice(t["offset"], t["offset"] + t["count"])
        if leaf in ("gamma", "moving_variance"):
            blob[sl] = 1.0
        elif leaf in ("beta", "moving_mean", "bias"):
            blob[sl] = 0.0
    return blob
