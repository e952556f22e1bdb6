"""Drop-in for multilingual_kws/embedding/transfer_learning.py on MI355X.

transfer_learn() keeps the reference's signature and return contract (name, model, details)
(reference :14-123): a frozen embedding (everything up to "dense_2") with a fresh
Dense(18,tanh) -> Dense(3,softmax) head trained with Adam on freshly augmented batches.
Each step runs augmentation -> micro-frontend -> EfficientNet-B0 forward -> head fwd/bwd/Adam as HIP
kernels; under torch.distributed the batch is sharded across ranks and the head gradient is
all-reduced over RCCL (multilingual_kws_amd/parallel.py).
"""
import csv
import glob
import json
import os
from typing import Dict, List, Optional

import numpy as np

from .. import parallel, weights
from ..embedding_model import EmbeddingModel
from ..head import Head, glorot_uniform_params
from . import input_data

CATEGORIES = 3   # silence + unknown + target keyword


def load_base_model(base_model_path, max_batch=1024, base_model_output="dense_2"):
    """The frozen embedding model.  base_model_path: what the reference passes to tf.keras.models.load_model
    (transfer_learning.py:36) -- a Keras SavedModel directory such as multilingual_context_73_0.8011, read without
    TensorFlow by multilingual_kws_amd.checkpoint_import -- or a Keras `.h5` file (model.save / save_weights; read by the same module's
    pure-Python HDF5 reader) -- or a weight-container directory written by
    multilingual_kws_amd.weights.save() (flat float32 blob + manifest with Keras variable names), or
    "synthetic[:SEED]" for the seeded random weights used by benchmarks."""
    p = str(base_model_path)
    if p.startswith("synthetic"):
        seed = int(p.split(":")[1]) if ":" in p else weights.DEFAULT_SEED
        blob = weights.synthetic_blob(seed)
    elif os.path.exists(os.path.join(p, "variables", "variables.index")):
        from .. import checkpoint_import
        blob = checkpoint_import.import_savedmodel(p)
    elif os.path.isfile(p) and p.lower().endswith((".h5", ".hdf5", ".keras.h5")):
        from .. import checkpoint_import                # the other format tf.keras.models.load_model takes: read without libhdf5
        blob = checkpoint_import.import_h5(p)
    else:
        blob = weights.load(p)
    return EmbeddingModel(blob, max_batch=max_batch, output=base_model_output), blob


class TransferLearnedModel:
    """What transfer_learn returns as `model`: frozen embedding + few-shot head with the two Keras
    methods the reference's callers use: predict (run.py, evaluate_files_*) and save (run.py:300)."""

    name = "TransferLearnedModel"

    def __init__(self, embedding, head, blob=None, base_model_path=None):
        self.embedding, self.head = embedding, head
        self._blob, self.base_model_path = blob, base_model_path
        self.history = None

    def predict_device(self, spec):
        """CUDA [N,49,40(,1)] -> CUDA [N,3] class probabilities."""
        import torch
        outs = []
        for s in range(0, spec.shape[0], self.embedding.max_batch):
            outs.append(self.head.forward(self.embedding.forward(spec[s:s + self.embedding.max_batch])))
        return torch.cat(outs) if outs else torch.empty((0, self.head.classes), device=self.embedding.device)

    def predict(self, x, batch_size=None, verbose=0):
        """numpy [N,49,40,1] (or [N,49,40]) -> numpy [N,3]."""
        import torch
        x = torch.as_tensor(np.asarray(x, dtype=np.float32)).to(self.embedding.device)
        if x.dim() == 4:
            x = x[..., 0]
        # checked(): a failed in-kernel exchange poisons the forward that ran it and is otherwise reported by the NEXT call (include/mkws.h);
        # the copy to the host has synchronised anyway, so the host API looks at the handle now and never returns the poisoned batch
        return self.embedding.checked(lambda: self.predict_device(x).cpu().numpy())

    def save(self, path):
        """Directory with head.npz (+ a copy of / pointer to the base weights)."""
        os.makedirs(path, exist_ok=True)
        np.savez(os.path.join(path, "head.npz"), params=self.head.get_params(),
                 dims=np.asarray([self.head.in_dim, self.head.hidden, self.head.classes]))
        meta = {"format": "mkws-transfer-learned-v1", "base_model_path": self.base_model_path, "base_model_output": self.embedding.output}
        if self._blob is not None and not str(self.base_model_path).startswith("synthetic"):
            weights.save(os.path.join(path, "base"), self._blob)
            meta["base_model_path"] = "base"
        with open(os.path.join(path, "model.json"), "w") as f:
            json.dump(meta, f)

    @classmethod
    def load(cls, path, max_batch=1024):
        meta = json.load(open(os.path.join(path, "model.json")))
        base = meta["base_model_path"]
        if not str(base).startswith("synthetic") and not os.path.isabs(base):
            base = os.path.join(path, base)
        emb, blob = load_base_model(base, max_batch, meta.get("base_model_output", "dense_2"))
        z = np.load(os.path.join(path, "head.npz"))
        i, h, c = [int(v) for v in z["dims"]]
        return cls(emb, Head(i, h, c, max_batch=max_batch, params=z["params"]), blob, meta["base_model_path"])


def load_models_shared(model_paths, max_batch=1024):
    """[TransferLearnedModel.save() directories] -> [TransferLearnedModel] that share ONE embedding handle wherever their base weights are
    the same bytes (the usual case: N few-shot heads on one multilingual embedding, run.py's `modelpaths`).  Models with other base
    weights get their own handle.  Multi-keyword serving (batch_streaming_analysis.streaming_inferences / multi_keyword_detections)
    runs one embedding pass per distinct handle."""
    import hashlib
    shared, out = {}, []
    for path in model_paths:
        path = os.fspath(path)
        meta = json.load(open(os.path.join(path, "model.json")))
        base = meta["base_model_path"]
        if not str(base).startswith("synthetic") and not os.path.isabs(base):
            base = os.path.join(path, base)
        cut = meta.get("base_model_output", "dense_2")
        if str(base).startswith("synthetic"):
            key = (str(base), cut)
        else:       # content, not location: every saved model carries its own copy of the base weights
            h = hashlib.sha256()
            if os.path.isdir(base):                 # SavedModel directory / blob directory
                kind = "dir"
                for root, _, files in sorted(os.walk(base)):
                    for f in sorted(files):
                        h.update(os.path.relpath(os.path.join(root, f), base).encode())
                        h.update(open(os.path.join(root, f), "rb").read())
            elif os.path.isfile(base):              # a Keras .h5 / .npy blob named by model.json (models saved without a blob of their own)
                kind = "file"
                h.update(open(base, "rb").read())
            else:
                raise FileNotFoundError(f"{path}: base model '{base}' (model.json: base_model_path) does not exist")
            key = (kind, h.hexdigest(), cut)
        if key not in shared:
            shared[key] = load_base_model(base, max_batch, cut)
        emb, blob = shared[key]
        z = np.load(os.path.join(path, "head.npz"))
        i, h_, c = [int(v) for v in z["dims"]]
        out.append(TransferLearnedModel(emb, Head(i, h_, c, max_batch=max_batch, params=z["params"], device=emb.device), blob, meta["base_model_path"]))
    return out


FORWARD_CLIPS = 3072     # clips per embedding forward of the frozen phase.  Measured on one MI355X, 512 clips per optimizer step, same calls
                         # (profiles/r05_notes.md section 6): 512 clips per forward (a forward per step) 0.74 ms per step; 1024: 0.58; 2048: 0.548; 3072: 0.534;
                         # 4096: 0.536 (whole multiples of the 1024-clip plan: its workgroup loops run several rounds per launch -- fewer launch
                         # boundaries and tails per clip; 1536 / 2560 clips, a partial last round, are slower than 1024 / 2048)
OVERLAP_FROM_GROUP = 4   # optimizer steps on a second stream from this many steps per forward: +3-4 % at 4 / 6 / 8 steps, -1 % at 2


def steps_per_forward(batch_size, forward_clips=None):
    """Optimizer steps whose batches share one forward pass of the frozen embedding."""
    return max(1, int(FORWARD_CLIPS if forward_clips is None else forward_clips) // max(int(batch_size), 1))


class FrozenHeadTrainer:
    """The first phase of transfer_learn (reference transfer_learning.py:38-93: `embedding.trainable = False`, head fitted with Adam)
    as a stream of optimizer steps that does not pay for small forward passes.

    The embedding is frozen, so the forward pass of batch t + 1 is independent of the head update of step t.  G = steps_per_forward
    consecutive batches are therefore drawn together (input_data.BatchGroups: the same draws in the same order as one batch at a time),
    augmented / featurised / SpecAugmented by ONE launch each and embedded by ONE forward pass over G * batch_size clips on a handle
    planned for that size; then the G optimizer steps run one after the other, each on its own batch_size rows and with its own
    all-reduce under torch.distributed -- the semantics of parallel.dp_step, unchanged.  With overlap=True the optimizer steps (a few
    small launches each, nowhere near filling the chip) go to a second stream and run under the NEXT group's augmentation, frontend and
    embedding kernels; two embedding buffers alternate, events order producer and consumer.  Measured on one MI355X at 512 clips per
    step (profiles/r05_notes.md): with two steps per forward the second stream loses 1 % (853.9 k -> 844.2 k clips/s: the head's few
    launches only take turns with the embedding's), from four steps per forward it gains 3-4 % (932 k -> 966 k at 4, 958 k -> 984 k at 6):
    overlap=None switches it on from OVERLAP_FROM_GROUP steps per forward.  Results are bit-identical either way.

    step() performs exactly one optimizer step and returns its [sum of row losses, #correct] (summed over ranks) as a device tensor
    that is rewritten by the next step() and lives on the trainer's stream: add it up with accumulate(), or read it after finish()
    (which joins the side stream; call it before reading head parameters elsewhere)."""

    def __init__(self, embedding, head, train_ds, batch_size, lr, group=None, overlap=None):
        import torch
        self.embedding, self.head, self.lr = embedding, head, lr
        self.bs = int(batch_size)
        self.G = max(1, min(int(group) if group is not None else steps_per_forward(self.bs), embedding.max_batch // self.bs))
        self.groups = train_ds if isinstance(train_ds, input_data.BatchGroups) else input_data.BatchGroups(train_ds)
        if self.groups.bs != self.bs:
            raise ValueError(f"dataset is batched by {self.groups.bs}, trainer by {self.bs}")
        self.device = embedding.device
        self.overlap = (self.G >= OVERLAP_FROM_GROUP) if overlap is None else bool(overlap)
        self.side = torch.cuda.Stream(device=self.device) if self.overlap else None
        nbuf = 2 if self.overlap else 1
        self.emb = [torch.empty((self.G * self.bs, embedding.output_dim), dtype=torch.float32, device=self.device) for _ in range(nbuf)]
        self.consumed = [None] * nbuf            # event: every optimizer step that reads emb[k] has run
        self.k, self.j, self.g, self.cur = -1, 0, 0, None
        self.forwards = 0

    def _refill(self, g):
        """Next group of g batches: draws + one launch chain on the caller's stream, hand-over to the optimizer stream."""
        import torch
        self.k = (self.k + 1) % len(self.emb)
        main = torch.cuda.current_stream(self.device)
        spec, labels = self.groups.take(g)
        labels = labels.to(torch.int32)
        if self.consumed[self.k] is not None:
            main.wait_event(self.consumed[self.k])           # the steps of two groups ago have finished with this buffer
        emb = self.embedding.forward(spec, out=self.emb[self.k][:g * self.bs])
        self.forwards += 1
        if self.overlap:
            ready = torch.cuda.Event()
            ready.record(main)
            self.side.wait_event(ready)
            labels.record_stream(self.side)
        self.cur, self.g, self.j = (emb, labels), g, 0

    def step(self, group_limit=None):
        """One optimizer step.  group_limit: steps left before the caller needs the head (end of an epoch): a new group is cut to it."""
        import torch
        from .. import parallel
        if self.j >= self.g:
            self._refill(self.G if group_limit is None else max(1, min(self.G, int(group_limit))))
        emb, labels = self.cur
        lo, hi = self.j * self.bs, (self.j + 1) * self.bs
        self.j += 1
        if not self.overlap:
            return parallel.dp_step(self.head, emb[lo:hi], labels[lo:hi], lr=self.lr)
        with torch.cuda.stream(self.side):
            stats = parallel.dp_step(self.head, emb[lo:hi], labels[lo:hi], lr=self.lr)
            if self.j >= self.g:
                ev = self.consumed[self.k] = self.consumed[self.k] or torch.cuda.Event()
                ev.record(self.side)
        return stats

    def accumulate(self, acc, stats):
        """acc += stats on the stream the optimizer steps run on (the statistics tensor is rewritten by the next step)."""
        import torch
        if self.overlap:
            with torch.cuda.stream(self.side):
                acc += stats.to(acc.dtype)
        else:
            acc += stats.to(acc.dtype)

    def finish(self):
        """The caller's stream waits for every optimizer step issued so far."""
        import torch
        if self.overlap:
            torch.cuda.current_stream(self.device).wait_stream(self.side)


def transfer_learn(
    target,
    train_files,
    val_files,
    unknown_files,
    num_epochs,
    num_batches,
    batch_size,
    primary_lr,
    backprop_into_embedding,
    embedding_lr,
    model_settings: Dict,
    base_model_path: os.PathLike,
    base_model_output: str,
    UNKNOWN_PERCENTAGE: float = 50.0,
    bg_datadir: os.PathLike = "/home/mark/tinyspeech_harvard/speech_commands/_background_noise_/",
    csvlog_dest: Optional[os.PathLike] = None,
    verbose=1,
    seed=None,
):
    """Single-target few-shot fine-tune; see the reference's docstring ("this only works for
    single-target models").  Extra keyword: `seed` (augmentation + head init; rank is added under DP).
    batch_size is the PER-RANK batch under torch.distributed (weak scaling)."""
    from ..embedding_model import OUTPUT_LAYERS
    if base_model_output not in OUTPUT_LAYERS:       # reference :38-42 cuts at get_layer(name=base_model_output)
        raise ValueError(f"base_model_output {base_model_output!r}: this build cuts the embedding at one of {sorted(OUTPUT_LAYERS)}")
    if backprop_into_embedding and base_model_output != "dense_2":
        raise ValueError('backprop_into_embedding=True is implemented for base_model_output="dense_2" (the only value the reference\'s callers pass)')
    import torch
    rank, world = parallel.rank(), parallel.world_size()
    group = steps_per_forward(batch_size)                # optimizer steps per forward pass of the frozen embedding (FrozenHeadTrainer)
    embedding, blob = load_base_model(base_model_path, max_batch=max(batch_size * group, 64), base_model_output=base_model_output)
    feat = embedding.output_dim
    head_seed = None if seed is None else int(seed)
    p0 = glorot_uniform_params(feat, 18, CATEGORIES, head_seed)
    if world > 1:      # identical initial head on every rank
        t = torch.from_numpy(p0).to(embedding.device)
        parallel.broadcast_(t, 0)
        p0 = t.cpu().numpy()
    head = Head(feat, 18, CATEGORIES, max_batch=max(batch_size, 64), params=p0, device=embedding.device)
    xfer = TransferLearnedModel(embedding, head, blob, str(base_model_path))

    audio_dataset = input_data.AudioDataset(
        model_settings=model_settings,
        commands=[target],
        background_data_dir=bg_datadir,
        unknown_files=unknown_files,
        unknown_percentage=UNKNOWN_PERCENTAGE,
        spec_aug_params=input_data.SpecAugParams(percentage=80),
        seed=None if seed is None else int(seed) + 1000003 * rank,   # decorrelated per-rank streams
    )
    AUTOTUNE = input_data.AUTOTUNE
    init_train_ds = audio_dataset.init_single_target(AUTOTUNE, train_files, is_training=True)
    init_val_ds = audio_dataset.init_single_target(AUTOTUNE, val_files, is_training=False)
    train_ds = init_train_ds.shuffle(buffer_size=1000).repeat().batch(batch_size)
    val_ds = init_val_ds.batch(batch_size)

    steps_per_epoch = batch_size * num_batches          # (sic) -- reference :89
    train_groups = input_data.BatchGroups(train_ds)      # ONE stream of batches for both phases (the draws continue across them)
    frozen = FrozenHeadTrainer(embedding, head, train_groups, batch_size, primary_lr, group=group)
    phases = [("head", primary_lr)]
    if backprop_into_embedding:
        # reference :94-112: `layer.trainable = True` on the nested embedding Model un-freezes ALL of its layers
        # (BatchNormalization too), the model is re-compiled with a fresh Adam(embedding_lr) and fitted again for
        # the same number of epochs; the returned history / val_accuracy are the second fit's.
        phases.append(("all", embedding_lr))
    trainer, history = None, None
    for phase, lr in phases:
        if phase == "all":
            from ..arch import BLOCKS
            from ..embedding_trainer import DROP_CONNECT_RATE, EmbeddingTrainer
            trainer = EmbeddingTrainer(blob, device=embedding.device)
            head.reset_optimizer()
            graphed = None          # one hipGraph replay per step (single process; the DP phase keeps the launch-by-launch path)
        history = {"loss": [], "accuracy": [], "val_loss": [], "val_accuracy": []}
        for epoch in range(num_epochs):
            acc_stats = torch.zeros(2, dtype=torch.float64, device=embedding.device)
            seen = 0
            for step_i in range(steps_per_epoch):
                if trainer is None:
                    # frozen embedding: G batches per forward pass, never across the end of an epoch (validation reads the head there)
                    stats = frozen.step(group_limit=steps_per_epoch - step_i)
                    frozen.accumulate(acc_stats, stats)
                    seen += batch_size * world
                    continue
                spec, labels = train_groups.take(1)
                if world == 1:
                    nb = spec.shape[0]
                    masks = {name: audio_dataset.rng.uniform(0, 1, nb) >= DROP_CONNECT_RATE * bi / len(BLOCKS)
                             for bi, (name, cin, cout, k, s, e) in enumerate(BLOCKS) if s == 1 and cin == cout}
                    if graphed is None or graphed.B != nb:
                        from ..embedding_trainer import TrainStepGraph
                        graphed = TrainStepGraph(trainer, head, nb, lr)
                    stats = graphed.run(spec, labels, masks)
                else:
                    nb = spec.shape[0]
                    masks = {name: audio_dataset.rng.uniform(0, 1, nb) >= DROP_CONNECT_RATE * bi / len(BLOCKS)
                             for bi, (name, cin, cout, k, s, e) in enumerate(BLOCKS) if s == 1 and cin == cout}
                    emb = trainer.forward_train(spec, masks)
                    stats = head.loss_grad(emb, labels)
                    trainer.backward(head.input_grad(nb), allreduce=world > 1)
                    if world > 1:
                        stats = parallel.allreduce_sum_(head.grad_view(with_stats=True))[-2:]
                    head.adam_step(lr=lr, grad_scale=1.0 / world)
                    trainer.adam_step(lr=lr, grad_scale=1.0 / world)
                acc_stats += stats.to(torch.float64)
                seen += spec.shape[0] * world
            frozen.finish()
            tl, ta = (acc_stats / max(seen, 1)).tolist()
            if trainer is not None:      # validation runs the inference kernels on the current weights (moving statistics)
                blob = trainer.blob()
                embedding = EmbeddingModel(blob, max_batch=max(batch_size, 64), device=embedding.device, output=base_model_output)
                xfer.embedding, xfer._blob = embedding, blob
                xfer.base_model_path = "fine-tuned:" + str(base_model_path)
            # validation: every rank evaluates the full (small) validation set
            vstats, vseen = np.zeros(2), 0
            for spec, labels in val_ds:
                probs = xfer.predict_device(spec[..., 0])
                lab = labels.long()
                vstats[0] += float(-torch.log(torch.clamp(probs[torch.arange(len(lab)), lab], min=1e-7)).sum())
                vstats[1] += float((probs.argmax(1) == lab).sum())
                vseen += len(lab)
            vl, va = (vstats / max(vseen, 1)).tolist()
            for k, v in zip(("loss", "accuracy", "val_loss", "val_accuracy"), (tl, ta, vl, va)):
                history[k].append(v)
            if verbose and rank == 0:
                print(f"Epoch {epoch + 1}/{num_epochs} - {steps_per_epoch} steps - loss: {tl:.4f} - accuracy: {ta:.4f} "
                      f"- val_loss: {vl:.4f} - val_accuracy: {va:.4f}")
    if csvlog_dest is not None and rank == 0:
        with open(csvlog_dest, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["epoch", "accuracy", "loss", "val_accuracy", "val_loss"])
            for e in range(num_epochs):
                w.writerow([e, history["accuracy"][e], history["loss"][e], history["val_accuracy"][e], history["val_loss"][e]])
    xfer.history = history

    va = history["val_accuracy"][-1] if history["val_accuracy"] else 0.0
    name = f"xfer_epochs_{num_epochs}_bs_{batch_size}_nbs_{num_batches}_val_acc_{va:0.2f}_target_{target}"
    details = dict(num_epochs=num_epochs, batch_size=batch_size, num_batches=num_batches, val_accuracy=va, target=target)
    return name, xfer, details


def _specs_for_files(files, model_settings):
    """file2spec over a list, batched: decode on the host, one frontend launch for all clips."""
    import torch
    n = model_settings["desired_samples"]
    audio = np.stack([input_data._read_wav(f, n) for f in files]) if len(files) else np.zeros((0, n), np.float32)
    if len(files) == 0:
        return np.zeros((0, model_settings["spectrogram_length"], model_settings["fingerprint_width"]), np.float32)
    return input_data.to_micro_spectrogram(model_settings, torch.from_numpy(audio).cuda()).cpu().numpy()


def _pick(words_to_evaluate, data_dir, utterances_per_word):
    fs_all = []
    for word in words_to_evaluate:
        wavs = glob.glob(data_dir + word + "/*.wav")
        if len(wavs) > utterances_per_word:
            fs = np.random.choice(wavs, utterances_per_word, replace=False)
        else:
            print("using all wavs for ", word)
            fs = wavs
        fs_all.extend(list(fs))
    return fs_all


def _split_confidences(preds, target_id):
    correct, incorrect = [], []
    for row, col in enumerate(np.argmax(preds, axis=1)):
        (correct if col == target_id else incorrect).append(preds[row][col])
    return dict(correct=correct, incorrect=incorrect)


def evaluate_fast_multiclass(words_to_evaluate: List[str], target_id: int, data_dir: os.PathLike,
                             utterances_per_word: int, model, model_settings: Dict):
    specs = _specs_for_files(_pick(words_to_evaluate, data_dir, utterances_per_word), model_settings)
    return _split_confidences(model.predict(np.expand_dims(specs, -1)), target_id)


def evaluate_fast_single_target(words_to_evaluate: List[str], target_id: int, data_dir: os.PathLike,
                                utterances_per_word: int, model, model_settings: Dict):
    specs = _specs_for_files(_pick(words_to_evaluate, data_dir, utterances_per_word), model_settings)
    preds = model.predict(np.expand_dims(specs, -1))
    return preds[:, target_id], preds


def evaluate_files_multiclass(files_to_evaluate: List[os.PathLike], target_id: int, model, model_settings: Dict):
    specs = _specs_for_files(files_to_evaluate, model_settings)
    return _split_confidences(model.predict(np.expand_dims(specs, -1)), target_id)


def evaluate_files_single_target(files_to_evaluate: List[os.PathLike], target_id: int, model, model_settings: Dict):
    """-> (preds[:, target_id], preds) exactly as the reference (:264-273)."""
    specs = _specs_for_files(files_to_evaluate, model_settings)
    preds = model.predict(np.expand_dims(specs, -1))
    return preds[:, target_id], preds
