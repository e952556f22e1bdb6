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
        return self.predict_device(x).cpu().numpy()

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
    embedding, blob = load_base_model(base_model_path, max_batch=max(batch_size, 64), base_model_output=base_model_output)
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
    train_iter = iter(train_ds)
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
            for _ in range(steps_per_epoch):
                spec, labels = next(train_iter)
                if trainer is None:
                    emb = embedding.forward(spec)
                    stats = parallel.dp_step(head, emb, labels, lr=lr)
                elif world == 1:
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
